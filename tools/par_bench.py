"""Micro-benchmark of the PAR kernels (HBM-bound): algorithmic GB/s per iteration launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from weclip_vit_comer_amd import synth
from weclip_vit_comer_amd.WeCLIP_model.PAR import PAR

B, C, H, W = int(os.environ.get("B", 16)), 3, 512, 512
img = synth.make_images(B, H, W).cuda()
masks = torch.rand(B, C, H, W, device="cuda")
mod = PAR([1, 2, 4, 8, 12, 24], 20).cuda()
for _ in range(3):
    out = mod(img, masks)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 10
e0.record()
for _ in range(n):
    out = mod(img, masks)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
alg = B * ((48 + 2 * C) * H * W * 4 * 20 + (3 + 48) * H * W * 4)
print(f"PAR forward B={B}: {ms:.3f} ms  -> {alg / ms / 1e6:.1f} GB/s algorithmic, {B / ms * 1e3:.0f} img/s")
