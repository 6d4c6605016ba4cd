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
