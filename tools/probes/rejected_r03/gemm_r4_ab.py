"""The 4-wave register-resident-fragments kernel (wc_gemm_set_r4) against the default 256x256 kernel: time and deviation."""
import ctypes, os, sys
os.environ.setdefault("WECLIP_GEMM_P192", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from weclip_vit_comer_amd import ops, _lib as L

lib = L.lib().cdll
lib.wc_gemm_set_r4.argtypes = [ctypes.c_int]
lib.wc_gemm_set_r4.restype = None


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


Mb = 16 * 1025
cases = [("8192^3", 8192, 8192, 8192, dict(o16=True)), ("qkv", Mb, 2304, 768, dict(o16=True)), ("proj", Mb, 768, 768, dict(o32=True, resid=True, round16=True)),
         ("fc1", Mb, 3072, 768, dict(o16=True, act=1)), ("fc2", Mb, 768, 3072, dict(o32=True, o16=True, resid=True)),
         ("fc1-bwd", Mb, 768, 3072, dict(o16=True)), ("k128", Mb, 1024, 128, dict(o16=True)), ("k64x3", Mb, 768, 192, dict(o16=True))]
for name, M, N, K, kw in cases:
    a = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") * 0.05).half()
    bias = torch.randn(N, device="cuda")
    res = torch.randn(M, N, device="cuda") if kw.get("resid") else None
    o16 = torch.empty(M, N, device="cuda", dtype=torch.float16) if kw.get("o16") else None
    o32 = torch.empty(M, N, device="cuda") if kw.get("o32") else None
    f = lambda: ops.gemm(a, w, M, N, K, bias=bias, resid=res, out16=o16, out32=o32, act=kw.get("act", 0), round16=kw.get("round16", False))
    r, outs = {}, {}
    for mode in (0, 1, 2, 0, 1, 2):
        lib.wc_gemm_set_r4(mode)
        o = o32 if o32 is not None else o16
        o.zero_()
        r.setdefault(mode, []).append(t(f))
        outs[mode] = o.clone()
    gf = 2.0 * M * N * K / 1e6
    a0, a1, a2 = min(r[0]), min(r[1]), min(r[2])
    print(f"{name:8s} M={M} N={N} K={K}: default {a0:7.1f} us ({gf / a0:6.1f} TF/s)   r4 {a1:7.1f} us ratio {a1 / a0:.3f} equal {bool(torch.equal(outs[0], outs[1]))}   "
          f"r4 with buffer_load..lds {a2:7.1f} us ({gf / a2:6.1f} TF/s) ratio {a2 / a0:.3f} equal {bool(torch.equal(outs[0], outs[2]))}", flush=True)
lib.wc_gemm_set_r4(0)
