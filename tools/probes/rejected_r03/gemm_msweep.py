"""How the 256x256 kernel's time per K-tile depends on the number of active CUs (tiles <= 256: one round): is the K loop
bound per CU (latency x bytes in flight) or by an aggregate delivery rate?  WECLIP_GEMM_PP_MIN_TILES=1 keeps small grids on it."""
import os, sys
os.environ["WECLIP_GEMM_PP_MIN_TILES"] = "1"
os.environ["WECLIP_GEMM_P192"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from weclip_vit_comer_amd import ops

def t(f, n=20, rounds=3):
    best = 1e9
    for _ in range(rounds):
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best

for N, K in ((768, 3072), (768, 768), (2304, 768)):
    for mt in (8, 16, 32, 48, 64, 85):
        M = mt * 256
        a = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") * 0.05).half()
        o = torch.empty(M, N, device="cuda", dtype=torch.float16)
        us = t(lambda: ops.gemm(a, w, M, N, K, out16=o))
        tiles = mt * (N // 256)
        print(f"N={N} K={K} M={M}: {tiles:4d} tiles ({tiles / 256:.2f} rounds) {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF/s  "
              f"{us / (K // 64) / max(1, -(-tiles // 256)):.2f} us per K-tile and round", flush=True)
