import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from weclip_vit_comer_amd import ops
M, N, K = [int(x) for x in sys.argv[1:4]]
a = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") * 0.05).half()
o = torch.empty(M, N, device="cuda", dtype=torch.float16)
for _ in range(5):
    ops.gemm(a, w, M, N, K, out16=o)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.gemm(a, w, M, N, K, out16=o)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"M={M} N={N} K={K} dbg={os.environ.get('WECLIP_GEMM_DBG','0')} stream={os.environ.get('WECLIP_GEMM_STREAM','1')}: {ms*1e3:8.1f} us {2.0*M*N*K/ms/1e9:7.1f} TF/s")
