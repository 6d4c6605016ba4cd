"""Run one tall GEMM (256x256 kernel), one 128x128 GEMM and one weight-gradient GEMM (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from weclip_vit_comer_amd import ops
M = 16384
a = torch.randn(M, 768, device="cuda").half(); w = (torch.randn(2304, 768, device="cuda") * 0.05).half()
o = torch.empty(M, 2304, device="cuda", dtype=torch.float16)
a2 = torch.randn(M, 256, device="cuda").half(); w2 = (torch.randn(256, 256, device="cuda") * 0.05).half()
o2 = torch.empty(M, 256, device="cuda", dtype=torch.float16)
dy = torch.randn(M, 1024, device="cuda").half(); x = torch.randn(M, 256, device="cuda").half()
for _ in range(3):
    ops.gemm(a, w, M, 2304, 768, out16=o)
    ops.gemm(a2, w2, M, 256, 256, out16=o2)
    ops.wgrad_partials(dy, x, M, 1024, 256, slices=16)
torch.cuda.synchronize()
