"""Run the attention forward + head-mean kernels alone (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from weclip_vit_comer_amd import ops
B, L, H, DH = 16, 1025, 12, 64
qkv = (torch.randn(B * L, 3 * H * DH, device="cuda") * 0.5).half()
for _ in range(4):
    ops.attention(qkv, B, L, H, DH, want_mean=True)
torch.cuda.synchronize()
