import os, sys
sys.path.insert(0, "/root/repo")
import torch
from weclip_vit_comer_amd import ops
for (M, N, K) in [(16, 3072, 768), (32, 3072, 768), (16, 768, 3072)]:
    a = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") * 0.05).half()
    bias = torch.randn(N, device="cuda"); o16 = torch.empty(M, N, device="cuda", dtype=torch.float16)
    f = lambda: ops.gemm(a, w, M, N, K, bias=bias, out16=o16, act=1)
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): f()
    e1.record(); torch.cuda.synchronize()
    print(f"skinny={os.environ.get('WECLIP_GEMM_SKINNY','1')} M={M} N={N} K={K}: {e0.elapsed_time(e1)/200*1e3:.1f} us (back-to-back launches)", flush=True)
