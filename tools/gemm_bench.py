"""GEMM micro-benchmark on the encoder shapes (algorithmic TFLOP/s per launch), the kernel variants interleaved in ONE
process (wc_gemm_set_mode: 1 = 256x256 ping-pong kernel with four phases per K-tile, 2 = two phases per K-tile)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from weclip_vit_comer_amd import ops, _lib as L


def make(M, N, K, nseg=1, out16=True, resid=False, act=0):
    a = ops.Split(torch.randn(M, K, device="cuda").half(), torch.randn(M, K, device="cuda").half() if nseg > 1 else None)
    w = ops.Split((torch.randn(N, K, device="cuda") * 0.05).half(), (torch.randn(N, K, device="cuda") * 0.05).half() if nseg > 2 else None)
    bias = torch.randn(N, device="cuda")
    res = torch.randn(M, N, device="cuda") if resid else None
    o16 = torch.empty(M, N, device="cuda", dtype=torch.float16) if out16 else None
    o32 = None if out16 else torch.empty(M, N, device="cuda")
    f = lambda: ops.gemm(a, w, M, N, K, bias=bias, resid=res, out16=o16, out32=o32, act=act)
    return f, (o16 if out16 else o32)


def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def bench(M, N, K, rounds=3, **kw):
    f, out = make(M, N, K, **kw)
    res = {1: [], 2: []}
    outs = {}
    for r in range(rounds):
        for mode in (1, 2):
            L.lib().wc_gemm_set_mode(mode)
            res[mode].append(timeit(f))
            if r == 0:
                outs[mode] = out.float().clone()
    d = (outs[1] - outs[2]).abs().max().item()
    line = f"M={M} N={N} K={K} {kw}:"
    for mode in (1, 2):
        ms = min(res[mode])
        line += f"  mode{mode} {ms*1e3:7.1f} us {2.0*M*N*K/ms/1e9:7.1f} TF/s"
    print(line + f"   max|mode1-mode2| = {d:.3e}", flush=True)


B = int(os.environ.get("GB_BATCH", "16"))
M = B * 1025
bench(M, 2304, 768)
bench(M, 768, 768, out16=False, resid=True)
bench(M, 3072, 768, act=1)
bench(M, 768, 3072, out16=False, resid=True)
bench(M, 2304, 768, nseg=3)
bench(M, 3072, 768, nseg=2, act=1)
bench(8192, 8192, 8192)
bench(4096, 4096, 4096)
L.lib().wc_gemm_set_mode(2)
