"""Head-mean attention map kernel alone at the encoder shape (HIP events around back-to-back launches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from weclip_vit_comer_amd import ops, _lib as L
B, L_, H, DH = int(os.environ.get("AB_B", 16)), int(os.environ.get("AB_L", 1025)), 12, 64
qkv = (torch.randn(B * L_, 3 * H * DH, device="cuda") * 0.5).half()
o16, lse, mean = ops.attention(qkv, B, L_, H, DH, want_mean=True)
lib = L.lib()
for _ in range(3):
    lib.wc_attn_mean(L.ptr(qkv), L.ptr(lse), L.ptr(mean), B, L_, H, DH, L.stream())
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 30
e0.record()
for _ in range(n):
    lib.wc_attn_mean(L.ptr(qkv), L.ptr(lse), L.ptr(mean), B, L_, H, DH, L.stream())
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / n * 1e3
print(f"attn_mean abl={os.environ.get('WECLIP_MEAN_ABL', '0')} B={B} L={L_}: {us:7.1f} us  {2.0 * B * H * L_ * L_ * DH / us / 1e6:7.1f} TFLOP/s", flush=True)
