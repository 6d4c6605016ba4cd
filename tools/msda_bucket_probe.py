"""Deformable-attention backward on the two shapes of the inserts (for rocprofv3 --kernel-trace --stats): direction = argv[1]
('c': 5 376 queries on one 32 x 32 level; 'v': 1 024 queries on 64^2 + 32^2 + 16^2), B = 16, 8 heads x 32, 4 points."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from weclip_vit_comer_amd.WeCLIP_model.comer import ms_deform_attn_core
N, M, D, P = 16, 8, 32, 4
if sys.argv[1] == "c":
    shapes, Lq = [(32, 32)], 5376
else:
    shapes, Lq = [(64, 64), (32, 32), (16, 16)], 1024
S = sum(h * w for h, w in shapes)
g = torch.Generator().manual_seed(0)
value = torch.randn(N, S, M, D, generator=g).cuda().requires_grad_(True)
loc = torch.rand(N, Lq, M, len(shapes), P, 2, generator=g).cuda().requires_grad_(True)
attn = torch.softmax(torch.randn(N, Lq, M, len(shapes) * P, generator=g), -1).view(N, Lq, M, len(shapes), P).cuda().requires_grad_(True)
for _ in range(6):
    out = ms_deform_attn_core(value, shapes, loc, attn)
    out.sum().backward()
torch.cuda.synchronize()
