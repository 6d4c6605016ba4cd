"""One weight-gradient GEMM shape of the ViT-CoMer inserts, a few launches (for rocprofv3 --pmc passes): WG_SLICES slices."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from weclip_vit_comer_amd import ops
M, N, K = 86016, 256, 256
s = int(os.environ.get("WG_SLICES", "64"))
dy = (torch.randn(M, N, device="cuda") * 0.1).half(); x = torch.randn(M, K, device="cuda").half()
for _ in range(4):
    ops.wgrad_partials(dy, x, M, N, K, slices=s, bias=True)
torch.cuda.synchronize()
