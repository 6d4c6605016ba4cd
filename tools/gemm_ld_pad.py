"""Does the leading dimension of the GEMM operands matter (L2 channel mapping of a K-tile's 256 row lines)?  The same product with
rows padded by 0 / 64 / 128 / 192 halves (0 / 128 / 256 / 384 B)."""
import os, sys
os.environ.setdefault("WECLIP_GEMM_P192", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from weclip_vit_comer_amd import ops


def t(f, n=20, rounds=4):
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


for name, M, N, K in (("qkv", 16400, 2304, 768), ("proj", 16400, 768, 768), ("fc1", 16400, 3072, 768), ("fc2", 16400, 768, 3072), ("8192^3", 8192, 8192, 8192)):
    o = torch.empty(M, N, device="cuda", dtype=torch.float16)
    res = []
    for pad in (0, 64, 128, 192, 0, 64):
        a = torch.randn(M, K + pad, device="cuda").half(); w = (torch.randn(N, K + pad, device="cuda") * 0.05).half()
        us = t(lambda: ops.gemm(a, w, M, N, K, lda=K + pad, ldw=K + pad, out16=o))
        res.append(f"pad {pad:3d}: {us:7.1f} us")
    print(f"{name:7s} M={M} N={N} K={K}: " + "   ".join(res), flush=True)
