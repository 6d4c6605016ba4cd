"""Attention forward (+ head-mean map) micro-benchmark at the encoder shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from weclip_vit_comer_amd import ops
B, L, H, DH = 16, 1025, 12, 64
qkv = (torch.randn(B * L, 3 * H * DH, device="cuda") * 0.5).half()
for want_mean in (False, True):
    for _ in range(4):
        ops.attention(qkv, B, L, H, DH, want_mean=want_mean)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.attention(qkv, B, L, H, DH, want_mean=want_mean)
    e1.record(); torch.cuda.synchronize()
    print(f"attention B={B} L={L} H={H} dh={DH} want_mean={want_mean}: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us", flush=True)
