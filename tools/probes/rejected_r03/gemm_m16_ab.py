"""256x256 tall-GEMM kernel: v_mfma_f32_32x32x16_f16 vs v_mfma_f32_16x16x32_f16 (wc_gemm_set_m16), interleaved in one process
on the encoder shapes and 8192^3: HIP-event timed (min over rounds) + maximum deviation from an fp32 torch product."""
import ctypes, os, sys
os.environ.setdefault("WECLIP_GEMM_P192", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from weclip_vit_comer_amd import ops, _lib as L

lib = L.lib().cdll
lib.wc_gemm_set_m16.argtypes = [ctypes.c_int]
lib.wc_gemm_set_m16.restype = None
lib.wc_gemm_set_w4.argtypes = [ctypes.c_int]
lib.wc_gemm_set_w4.restype = None
lib.wc_gemm_set_ring10.argtypes = [ctypes.c_int]
lib.wc_gemm_set_ring10.restype = None


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


Mb = int(os.environ.get("GB_BATCH", "16")) * 1025
cases = [("qkv", Mb, 2304, 768, dict(o16=True)), ("proj", Mb, 768, 768, dict(o32=True, resid=True, round16=True)),
         ("fc1", Mb, 3072, 768, dict(o16=True, act=1)), ("fc2", Mb, 768, 3072, dict(o32=True, o16=True, resid=True)),
         ("fc1-bwd", Mb, 768, 3072, dict(o16=True)), ("8192^3", 8192, 8192, 8192, dict(o16=True))]
for name, M, N, K, kw in cases:
    a = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") * 0.05).half()
    bias = torch.randn(N, device="cuda")
    res = torch.randn(M, N, device="cuda") if kw.get("resid") else None
    o16 = torch.empty(M, N, device="cuda", dtype=torch.float16) if kw.get("o16") else None
    o32 = torch.empty(M, N, device="cuda") if kw.get("o32") else None
    f = lambda: ops.gemm(a, w, M, N, K, bias=bias, resid=res, out16=o16, out32=o32, act=kw.get("act", 0), round16=kw.get("round16", False))
    ref = a[:2048].float() @ w.float().t() + bias
    if kw.get("round16"): ref = ref.half().float()
    if kw.get("act") == 1: ref = ref * torch.sigmoid(1.702 * ref)
    if res is not None: ref = ref + res[:2048]
    r, err = {}, {}
    outs = {}
    for mode in (0, 1, 2, 3, 0, 1, 2, 3):
        lib.wc_gemm_set_m16(1 if mode == 1 else 0)
        lib.wc_gemm_set_w4(1 if mode == 2 else 0)
        lib.wc_gemm_set_ring10(1 if mode == 3 else 0)
        o = o32 if o32 is not None else o16
        o.zero_()
        r.setdefault(mode, []).append(t(f))
        out = (o32 if o32 is not None else o16)[:2048].float()
        err[mode] = float((out - ref).abs().max() / ref.abs().max())
        outs[mode] = (o32 if o32 is not None else o16).clone()
    gf = 2.0 * M * N * K / 1e6
    a0, a1, a2, a3 = min(r[0]), min(r[1]), min(r[2]), min(r[3])
    same = bool(torch.equal(outs[0], outs[3]))
    print(f"{name:8s} M={M} N={N} K={K}: 32x32x16 {a0:7.1f} us ({gf / a0:6.1f} TF/s, err {err[0]:.1e})   16x16x32 {a1:7.1f} us "
          f"({gf / a1:6.1f} TF/s, err {err[1]:.1e}) ratio {a1 / a0:.3f}   4-wave {a2:7.1f} us ({gf / a2:6.1f} TF/s, err {err[2]:.1e}) ratio {a2 / a0:.3f}   10-slot ring {a3:7.1f} us ({gf / a3:6.1f} TF/s, bit-identical {same}) ratio {a3 / a0:.3f}", flush=True)
lib.wc_gemm_set_m16(0)
lib.wc_gemm_set_w4(0)
lib.wc_gemm_set_ring10(0)
