"""Weight-gradient GEMM (gemm_km_kernel) micro-benchmark: us per launch and TFLOP/s over the shapes of the step, per slice count."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from weclip_vit_comer_amd import ops


def bench(M, N, K, slices, n=20, rounds=3):
    dy = (torch.randn(M, N, device="cuda") * 0.1).half()
    x = torch.randn(M, K, device="cuda").half()
    f = lambda: ops.wgrad_partials(dy, x, M, N, K, slices=slices, bias=True)
    best = 1e9
    for _ in range(rounds):
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    tiles = ops.wgrad_tiles(N, K)
    print(f"M={M:6d} N={N:4d} K={K:4d} slices={slices:4d} ({tiles * slices:4d} workgroups): {best*1e3:7.1f} us  {2.0*M*N*(K+1)/best/1e9:6.1f} TF/s",
          flush=True)


for M in (86016, 16384):
    for s in (32, 64, 128):
        bench(M, 256, 256, s)
bench(86016, 256, 128, 128)
bench(86016, 128, 256, 128)
bench(16384, 256, 768, 16)
bench(16384, 256, 768, 32)
bench(16384, 256, 2816, 8)
bench(16384, 256, 2816, 16)
