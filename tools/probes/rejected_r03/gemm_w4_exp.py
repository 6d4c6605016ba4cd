"""Timing experiments on the 4-wave GEMM kernel (library built with -DW4_EXPERIMENTS): parts of the K loop removed (results wrong)."""
import ctypes, os, sys
os.environ.setdefault("WECLIP_GEMM_P192", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from weclip_vit_comer_amd import ops, _lib as L
lib = L.lib().cdll
lib.wc_gemm_set_w4.argtypes = [ctypes.c_int]; lib.wc_gemm_set_w4.restype = None

def t(f, n=10, rounds=3):
    best = 1e9
    for _ in range(rounds):
        for _ in range(2): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best

names = {0: "8-wave LDS-DMA kernel", 1: "4-wave kernel", 2: "  - global loads", 4: "  - loads - LDS writes", 5: "  - barrier only", 8: "  - loads - writes - barrier",
         16: "  - everything but MFMA", 9: "  - fragment reads only"}
for M, N, K in ((8192, 8192, 8192), (16384, 2304, 768)):
    a = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") * 0.05).half()
    o = torch.empty(M, N, device="cuda", dtype=torch.float16)
    f = lambda: ops.gemm(a, w, M, N, K, out16=o)
    tiles = (M // 256) * (N // 256); rounds = -(-tiles // 256)
    for mode, nm in names.items():
        lib.wc_gemm_set_w4(mode)
        us = t(f)
        print(f"M={M} N={N} K={K} {nm:32s} {us:8.1f} us  {us / (K // 64) / rounds:.3f} us per K-tile and round", flush=True)
lib.wc_gemm_set_w4(0)
