"""Fixed cost vs K-loop cost of the 256x256 GEMM: time(K) at a fixed M x N (one or two full rounds of tiles)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from weclip_vit_comer_amd import ops

def t(M, N, K, n=30):
    a = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") * 0.05).half()
    o = torch.empty(M, N, device="cuda", dtype=torch.float16)
    for _ in range(3): ops.gemm(a, w, M, N, K, out16=o)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): ops.gemm(a, w, M, N, K, out16=o)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for M, N in ((16384, 1024), (16384, 2048), (16384, 3072)):
    prev = None
    for K in (64, 128, 256, 512, 768, 1536, 3072):
        us = t(M, N, K)
        rounds = (M // 256) * (N // 256) / 256.0
        extra = "" if prev is None else f"  d/Ktile/round {(us - prev[1]) / ((K - prev[0]) / 64) / rounds:6.3f} us"
        print(f"M={M} N={N} K={K:5d} rounds={rounds:.2f}: {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF/s{extra}", flush=True)
        prev = (K, us)
