"""Row-streaming GEMM (csrc/gemm_row.hip) against the tile kernels behind ops.gemm on the ViT-CoMer shapes: us per launch and
the HBM rate over the ALGORITHMIC bytes (A once + every output / side input once).
    python tools/gemm_row_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from weclip_vit_comer_amd import ops

torch.manual_seed(0)


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for M in (86016, 16384):
    for N, K in ((256, 256), (128, 256), (256, 128)):
        a = torch.randn(M, K, device="cuda").half()
        w = (torch.randn(N, K, device="cuda") * 0.05).half()
        bias = torch.randn(N, device="cuda")
        res = torch.randn(M, N, device="cuda")
        o32 = torch.empty(M, N, device="cuda")
        o16 = torch.empty(M, N, device="cuda", dtype=torch.float16)
        u = torch.randn(M, N, device="cuda")
        g0, b0 = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
        l0, l1 = torch.empty_like(o16), torch.empty_like(o16)
        cases = [
            ("f16 out", dict(out16=o16), M * K * 2 + M * N * 2),
            ("f32 out + resid", dict(resid=res, out32=o32), M * K * 2 + 2 * M * N * 4),
            ("f32+f16 out + resid", dict(resid=res, out32=o32, out16=o16), M * K * 2 + 2 * M * N * 4 + M * N * 2),
            ("gelu: pre32 + f16", dict(act=6, pre32=o32, out16=o16), M * K * 2 + M * N * 6),
            ("gelu': aux -> f16", dict(act=7, aux=u, ldaux=N, out16=o16), M * K * 2 + M * N * 6),
        ]
        for name, kw, nbytes in cases:
            kw2 = dict(kw)
            if "aux" in kw2:
                kw2["rpg"] = 1
            t_old = timeit(lambda: ops.gemm(a, w, M, N, K, bias=bias, **kw2))
            t_new = timeit(lambda: ops.gemm_row(a, w, M, N, K, bias=bias, **kw))
            print(f"M={M} N={N} K={K} {name:22s}: tile kernels {t_old:7.1f} us ({nbytes / t_old / 1e6:5.2f} TB/s)   row kernel {t_new:7.1f} us "
                  f"({nbytes / t_new / 1e6:5.2f} TB/s)", flush=True)
        if N == 256:
            nb = M * K * 2 + 2 * M * N * 4 + 2 * M * N * 2
            t_new = timeit(lambda: ops.gemm_row(a, w, M, N, K, bias=bias, resid=res, out32=o32, ln=[(g0, b0, l0), (g0, b0, l1)]))
            t_old = timeit(lambda: (ops.gemm(a, w, M, N, K, bias=bias, resid=res, out32=o32),
                                    ops.layernorm(o32, g0, b0, want32=False, want16=True), ops.layernorm(o32, g0, b0, want32=False, want16=True)))
            print(f"M={M} N={N} K={K} f32 out + resid + 2 LN    : GEMM + 2 LayerNorm launches {t_old:7.1f} us   row kernel, fused {t_new:7.1f} us "
                  f"({nb / t_new / 1e6:5.2f} TB/s)", flush=True)
