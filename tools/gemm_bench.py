"""GEMM micro-benchmark on the encoder shapes (algorithmic TFLOP/s per launch; min over rounds, HIP-event timed)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from weclip_vit_comer_amd import ops


def bench(M, N, K, nseg=1, out16=True, resid=False, act=0, n=20, rounds=3):
    a = ops.Split(torch.randn(M, K, device="cuda").half(), torch.randn(M, K, device="cuda").half() if nseg > 1 else None)
    w = ops.Split((torch.randn(N, K, device="cuda") * 0.05).half(), (torch.randn(N, K, device="cuda") * 0.05).half() if nseg > 2 else None)
    bias = torch.randn(N, device="cuda")
    res = torch.randn(M, N, device="cuda") if resid else None
    o16 = torch.empty(M, N, device="cuda", dtype=torch.float16) if out16 else None
    o32 = None if out16 else torch.empty(M, N, device="cuda")
    f = lambda: ops.gemm(a, w, M, N, K, bias=bias, resid=res, out16=o16, out32=o32, act=act)
    best = 1e9
    for _ in range(rounds):
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    print(f"M={M} N={N} K={K} nseg={nseg} out16={out16} resid={resid} act={act}: {best*1e3:8.1f} us  {2.0*M*N*K/best/1e9:7.1f} TF/s", flush=True)


M = int(os.environ.get("GB_BATCH", "16")) * 1025
bench(M, 2304, 768)
bench(M, 768, 768, out16=False, resid=True)
bench(M, 3072, 768, act=1)
bench(M, 768, 3072, out16=False, resid=True)
bench(8192, 8192, 8192)
