"""256x192 vs 256x256 tall-GEMM tile on the encoder shapes, interleaved in one process (HIP-event timed, min over rounds)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from weclip_vit_comer_amd import ops, _lib as L

lib = L.lib().cdll
lib.wc_gemm_set_p192.argtypes = [ctypes.c_int, ctypes.c_float]
lib.wc_gemm_set_p192.restype = None


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


M = int(os.environ.get("GB_BATCH", "16")) * 1025
for name, N, K, kw in (("qkv", 2304, 768, dict(o16=True)), ("proj", 768, 768, dict(o32=True, resid=True, round16=True)),
                       ("fc1", 3072, 768, dict(o16=True, act=1)), ("fc2", 768, 3072, dict(o32=True, o16=True, resid=True))):
    a = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") * 0.05).half()
    bias = torch.randn(N, device="cuda")
    res = torch.randn(M, N, device="cuda") if kw.get("resid") else None
    o16 = torch.empty(M, N, device="cuda", dtype=torch.float16) if kw.get("o16") else None
    o32 = torch.empty(M, N, device="cuda") if kw.get("o32") else None
    f = lambda: ops.gemm(a, w, M, N, K, bias=bias, resid=res, out16=o16, out32=o32, act=kw.get("act", 0), round16=kw.get("round16", False))
    r = {}
    for mode in (0, 2, 0, 2):
        lib.wc_gemm_set_p192(mode, 0.0)
        r.setdefault(mode, []).append(t(f))
    gf = 2.0 * M * N * K / 1e6
    a0, a2 = min(r[0]), min(r[2])
    print(f"{name:5s} M={M} N={N} K={K}: 256x256 {a0:7.1f} us ({gf / a0:6.1f} TF/s)   256x192 {a2:7.1f} us ({gf / a2:6.1f} TF/s)   ratio {a2 / a0:.3f}", flush=True)
lib.wc_gemm_set_p192(1, 0.0)
