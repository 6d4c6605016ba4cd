"""Typed Python wrappers over the C ABI (include/weclip_hip.h).  Tensors in, tensors out; every
wrapper allocates its outputs with torch (device memory + current stream only) and calls the
HIP library.  No arithmetic happens in torch here."""
import math

import torch

from . import _lib as L

F16 = torch.float16
F32 = torch.float32
LOG2E = 1.4426950408889634


class KernelTimer:
    """Per-kernel HIP-event timing of the dominant kernels (bench.py roofline leg).  The events are recorded by
    the C library at its own launch sites (csrc/core.hip wc_prof_*), one pair per kernel launch on the launch
    stream, under the kernel names rocprofv3 reports; `work` is the algorithmic work (flops or bytes) of a launch."""

    active = False

    @staticmethod
    def enable(stride=1):
        """Record one of every `stride` instrumented launches (clearing earlier records); 0 / False stops."""
        L.lib().cdll.wc_prof_enable(int(stride))
        KernelTimer.active = bool(stride)
        if not stride:
            L.lib().cdll.wc_prof_tag(None)

    @staticmethod
    def tag(name):
        """Group tag for the launches that follow (only while the timers run: one ctypes call otherwise saved)."""
        if KernelTimer.active:
            L.lib().cdll.wc_prof_tag(name)

    @staticmethod
    def summary():
        """kernel name -> dict(ms total of the recorded launches, launches, work, est_ms = ms scaled by the sampling weight of
        each record: the kernel's estimated total over the instrumented steps) after a device sync."""
        import ctypes
        buf = ctypes.create_string_buffer(1 << 16)
        L.lib().cdll.wc_prof_report(buf, len(buf))
        out = {}
        for line in buf.value.decode().splitlines():
            name, n, ms, work, est, nbytes = line.split("\t")
            out[name] = {"ms": float(ms), "launches": int(n), "work": float(work), "est_ms": float(est), "bytes": float(nbytes)}
        return out


class Split:
    """An fp32 tensor represented for the MFMA GEMM as fp16 `hi` (+ optional fp16 `lo` with
    x ~= hi + lo).  `lo is None` means single-pass fp16 precision."""
    __slots__ = ("hi", "lo")

    def __init__(self, hi, lo=None):
        self.hi, self.lo = hi, lo


def split_f16(x, with_lo=False):
    """fp32 tensor -> Split (hi = fp16(x), lo = fp16(x - hi))."""
    L.require_gpu()
    x = x.detach().float().contiguous()
    hi = torch.empty(x.shape, device=x.device, dtype=F16)
    lo = torch.empty(x.shape, device=x.device, dtype=F16) if with_lo else None
    L.lib().wc_split_f16(L.ptr(x, F32, "x"), L.ptr(hi), L.ptr(lo), x.numel(), L.stream())
    return Split(hi, lo)


def gemm(a, w, M, N, K, *, lda=None, ldw=None, bias=None, resid=None, ldr=None, sR=None,
         out32=None, out16=None, out16lo=None, ldc=None, act=0, round16=False, scale=1.0,
         scale_cols=0, batch=1, sA=0, sW=0, sC=0, pre32=None, aux=None, rowmap=None, rpg=0,
         ldaux=0, auxh=None, cscale=None, sCS=0, zdiv=None, sA2=0, sW2=0, sC2=0, sB2=0, sX2=0):
    """C = epilogue(A W^T).  `a`, `w`: Split (or fp16 tensors).  Segments accumulated:
    (a.hi,w.hi) [+ (a.lo,w.hi)] [+ (a.hi,w.lo)].  zdiv / s?2: grouped launch (wc_gemm_f16_grouped): batch index
    z -> group z // zdiv (second-level strides) and member z % zdiv (sA / sW / sC)."""
    a = a if isinstance(a, Split) else Split(a)
    w = w if isinstance(w, Split) else Split(w)
    segs = [(a.hi, w.hi)]
    if a.lo is not None:
        segs.append((a.lo, w.hi))
    if w.lo is not None:
        segs.append((a.hi, w.lo))
    lda = K if lda is None else lda
    ldw = K if ldw is None else ldw
    ldc = N if ldc is None else ldc
    ap = [L.ptr(s[0], F16, "A") for s in segs] + [None] * (3 - len(segs))
    wp = [L.ptr(s[1], F16, "W") for s in segs] + [None] * (3 - len(segs))
    ldr = ldc if ldr is None else ldr
    sR = sC if sR is None else sR
    L.lib().wc_gemm_f16_grouped(ap[0], ap[1], ap[2], wp[0], wp[1], wp[2], len(segs), M, N, K, lda, ldw,
                                batch, sA, sW, sC, L.ptr(bias, F32, "bias"), L.ptr(resid, F32, "resid"),
                                ldr, sR, L.ptr(out32, F32, "out32"), L.ptr(out16, F16, "out16"),
                                L.ptr(out16lo, F16, "out16lo"), ldc, act, 1 if round16 else 0,
                                float(scale), scale_cols, L.ptr(pre32, F32, "pre32"), L.ptr(aux, F32, "aux"),
                                L.ptr(rowmap, torch.int32, "rowmap"), rpg, ldaux, L.ptr(auxh, F16, "auxh"),
                                L.ptr(cscale, F32, "cscale"), sCS, batch if zdiv is None else zdiv, sA2, sW2, sC2, sB2,
                                sX2, L.stream())


def gemm_row_ok(M, N, K):
    """Shapes of the row-streaming kernel (csrc/gemm_row.hip): N <= 256, N % 4 == 0, K = 128 | 256."""
    return N <= 256 and N % 4 == 0 and K in (128, 256) and M > 0


def gemm_row(a16, w16, M, N, K, *, lda=None, ldw=None, bias=None, cscale=None, act=0, aux=None, auxh=None, ldaux=0, resid=None,
             ldr=None, out32=None, out16=None, pre32=None, ldc=None, ldc16=None, ln=(), eps=1e-5):
    """C = epilogue(A W^T) on the row-streaming kernel; `ln`: up to two (gamma, beta, out16 (M, 256)) fused LayerNorm outputs."""
    lda = K if lda is None else lda
    ldw = K if ldw is None else ldw
    ldc = N if ldc is None else ldc
    ldr = ldc if ldr is None else ldr
    ldc16 = ldc if ldc16 is None else ldc16
    lnp = [None] * 6
    for i, (gam, bet, out) in enumerate(ln):
        lnp[3 * i:3 * i + 3] = [L.ptr(gam, F32, "ln.weight"), L.ptr(bet, F32, "ln.bias"), L.ptr(out, F16, "ln.out")]
    L.lib().wc_gemm_row_f16(L.ptr(a16, F16, "A"), lda, L.ptr(w16, F16, "W"), ldw, M, N, K, L.ptr(bias, F32, "bias"),
                            L.ptr(cscale, F32, "cscale"), act, L.ptr(aux, F32, "aux"), L.ptr(auxh, F16, "auxh"), ldaux,
                            L.ptr(resid, F32, "resid"), ldr, L.ptr(out32, F32, "out32"), L.ptr(out16, F16, "out16"),
                            L.ptr(pre32, F32, "pre32"), ldc, ldc16, lnp[0], lnp[1], lnp[2], lnp[3], lnp[4], lnp[5], float(eps), L.stream())


def gemm_row_grouped(a16, w16, M, N, K, groups, *, lda=None, ldw=None, bias=None, act=0, auxh=None, ldaux=0, out32=None, out16=None,
                     ldc=None, ldc16=None, gA=0, gW=0, gB=0, gC=0, gX=0):
    """`groups` products of one shape on the row-streaming kernel: group i reads a16 + i*gA, w16 + i*gW, bias + i*gB, auxh + i*gX
    and writes out32 / out16 + i*gC (element offsets)."""
    lda = K if lda is None else lda
    ldw = K if ldw is None else ldw
    ldc = N if ldc is None else ldc
    ldc16 = ldc if ldc16 is None else ldc16
    L.lib().wc_gemm_row_f16_grouped(L.ptr(a16, F16, "A"), lda, L.ptr(w16, F16, "W"), ldw, M, N, K, L.ptr(bias, F32, "bias"), act,
                                    L.ptr(auxh, F16, "auxh"), ldaux, L.ptr(out32, F32, "out32"), L.ptr(out16, F16, "out16"), ldc, ldc16,
                                    groups, gA, gW, gB, gC, gX, L.stream())


def layernorm(x, weight, bias, *, eps=1e-5, want32=False, want16=True, with_lo=False, rows=None,
              D=None, ldx=None):
    """Row LayerNorm of fp32 x (rows, D).  Returns (y32 or None, Split or None)."""
    D = x.shape[-1] if D is None else D
    rows = x.numel() // D if rows is None else rows
    ldx = D if ldx is None else ldx
    dev = x.device
    y32 = torch.empty(rows, D, device=dev, dtype=F32) if want32 else None
    hi = torch.empty(rows, D, device=dev, dtype=F16) if want16 else None
    lo = torch.empty(rows, D, device=dev, dtype=F16) if (want16 and with_lo) else None
    L.lib().wc_layernorm(L.ptr(x, F32, "x"), ldx, L.ptr(weight, F32, "ln.weight"),
                         L.ptr(bias, F32, "ln.bias"), eps, L.ptr(y32), L.ptr(hi), L.ptr(lo), rows, D,
                         L.stream())
    return y32, (Split(hi, lo) if want16 else None)


def attention(qkv16, B, Lq, H, DH, want_mean=True, want_o32=False):
    """qkv16 (B*L, 3E) fp16 with q pre-scaled by log2(e)/sqrt(DH).
    Returns (o16 (B*L, E) fp16, lse (B,H,L) f32, mean (B,L,L) f32 or None)."""
    E = H * DH
    dev = qkv16.device
    lib = L.lib()
    o16 = torch.empty(B * Lq, E, device=dev, dtype=F16)
    lse = torch.empty(B, H, Lq, device=dev, dtype=F32)
    o32 = torch.empty(B * Lq, E, device=dev, dtype=F32) if want_o32 else None
    lib.wc_attn_fwd(L.ptr(qkv16, F16, "qkv"), L.ptr(o16), L.ptr(o32), L.ptr(lse), B, Lq, H, DH, L.stream())
    mean = None
    if want_mean:
        mean = torch.empty(B, Lq, Lq, device=dev, dtype=F32)
        lib.wc_attn_mean(L.ptr(qkv16), L.ptr(lse), L.ptr(mean), B, Lq, H, DH, L.stream())
    if want_o32:
        return o16, lse, mean, o32
    return o16, lse, mean


def q_scale(DH):
    return LOG2E / math.sqrt(DH)


# ------------------------------------------------------------------------------------------------
# backward helpers (csrc/train_ops.hip, csrc/attention_bwd.hip)

def attention_bwd(qkv16, do16, o32, lse, B, Lq, H, DH, with_lo=True):
    """-> dqkv Split (B*L, 3E): gradient w.r.t. the unscaled in-projection output."""
    E = H * DH
    dev = qkv16.device
    Lp = (Lq + 63) // 64 * 64
    ws = torch.empty(3, B * H * DH * Lp, device=dev, dtype=F16)
    delta = torch.empty(B * H * Lq, device=dev, dtype=F32)
    d = Split(torch.empty(B * Lq, 3 * E, device=dev, dtype=F16),
              torch.empty(B * Lq, 3 * E, device=dev, dtype=F16) if with_lo else None)
    L.lib().wc_attn_bwd(L.ptr(qkv16, F16, "qkv"), L.ptr(do16, F16, "dO"), L.ptr(o32, F32, "o32"), L.ptr(lse, F32, "lse"),
                        L.ptr(ws[0]), L.ptr(ws[1]), L.ptr(ws[2]), L.ptr(delta), L.ptr(d.hi), L.ptr(d.lo), B, Lq, Lp,
                        H, DH, L.stream())
    return d


def transpose_f16(src, R, C, *, ld=None, batch=1, sSrc=0, with_lo=False, scale=1.0, oR=None, ones_row=False):
    """out (C, Kp) fp16 with out[c, b*R + r] = scale*src[b, r, c]; Kp = batch*R rounded up to 64
    (zero padded) so it can be the K dimension of a weight-gradient GEMM."""
    ld = C if ld is None else ld
    oR = R if oR is None else oR          # output column stride between batches (>= R, zero padded)
    K = (batch - 1) * oR + R
    Kp = (batch * oR + 63) // 64 * 64
    dev = src.device
    alloc = torch.zeros if Kp != K else torch.empty      # only the K padding needs zeros
    rows = C + 1 if ones_row else C     # optional extra row of ones: dY^T [X | 1] yields the bias gradient too
    hi = alloc(rows, Kp, device=dev, dtype=F16)
    lo = alloc(rows, Kp, device=dev, dtype=F16) if with_lo else None
    if ones_row:
        if oR != R:
            raise RuntimeError("ones_row needs densely packed batches")
        hi[C, :K] = 1.0
        if lo is not None:
            lo[C].zero_()
    f32 = src.dtype == F32
    L.lib().wc_transpose_f16(L.ptr(src, F32 if f32 else F16, "src"), 1 if f32 else 0, ld, sSrc, L.ptr(hi), L.ptr(lo),
                             Kp, oR, batch, R, C, float(scale), L.stream())
    return Split(hi, lo), Kp


_ZEROS = {}


def wgrad_tiles(N, K, bias=True):
    """128x128 output tiles of a weight-gradient GEMM (csrc/gemm_km.hip): the bias column K takes a column tile of its own only
    when K is not a multiple of 128 (otherwise the last column tile's waves store it)."""
    kt = K // 128 if (bias and K % 128 == 0) else (K + (1 if bias else 0) + 127) // 128
    return ((N + 127) // 128) * kt


def wgrad_partials(dy16, x16, M, N, K, *, lda=None, ldx=None, slices=1, bias=True, xmap=None, groups=1, gA=0, gX=0):
    """Split-K partials of dW = dY^T X (and db = dY^T 1 as column K) from ROW-MAJOR fp16 operands
    dy16 (M, lda), x16 (rows, ldx): -> (part (ns, N, K + bias) fp32, ns).  xmap = (rows_per_group, group_stride,
    offset): token m reads X row (m // rpg) * stride + m % rpg + offset (patch rows of a (B, 1 + hw, C) tensor).
    groups > 1: that many gradients of one shape in one launch, group i reading dy16 + i*gA and x16 + i*gX (elements);
    part is then (groups, ns, N, K + bias)."""
    L.require_gpu()
    lda = N if lda is None else lda
    ldx = K if ldx is None else ldx
    dev = dy16.device
    z = _ZEROS.get(dev)
    if z is None:
        z = _ZEROS[dev] = torch.zeros(64, device=dev, dtype=F16)
    mslice = (-(-M // slices) + 63) // 64 * 64
    ns = -(-M // mslice)
    K1 = K + (1 if bias else 0)
    part = torch.empty((ns, N, K1) if groups == 1 else (groups, ns, N, K1), device=dev, dtype=F32)
    rpg, gs, off = xmap if xmap is not None else (max(M, 64), 0, 0)
    L.lib().wc_gemm_km_f16_grouped(L.ptr(dy16, F16, "dY"), lda, L.ptr(x16, F16, "X"), ldx, L.ptr(z), M, N, K, rpg, gs, off,
                                   mslice, 1 if bias else 0, L.ptr(part), groups, gA, gX, L.stream())
    return part, ns


def colsum(src, R, C, *, ld=None, alpha=1.0, round16=False, out=None):
    ld = C if ld is None else ld
    dev = src.device
    part = torch.empty(((R + 255) // 256) * C, device=dev, dtype=F32)
    out = torch.empty(C, device=dev, dtype=F32) if out is None else out
    f32 = src.dtype == F32
    L.lib().wc_colsum(L.ptr(src, F32 if f32 else F16, "src"), 1 if f32 else 0, ld, L.ptr(part), L.ptr(out, F32), R, C,
                      float(alpha), 1 if round16 else 0, L.stream())
    return out


def layernorm_bwd(dy, x, w, *, add=None, want32=True, want16=False, out_scale=1.0, alpha=1.0, eps=1e-5, dgb=None):
    """-> (dx32 or None, dx16 or None, dgb (2, D) = alpha*[dgamma; dbeta]); `dgb`, if given, is the (2, D) f32
    destination (e.g. the adjacent weight / bias gradient views of a flat gradient bucket)."""
    rows, D = x.shape
    dev = x.device
    dx32 = torch.empty(rows, D, device=dev, dtype=F32) if want32 else None
    dx16 = torch.empty(rows, D, device=dev, dtype=F16) if want16 else None
    part = torch.empty(((rows + 15) // 16) * 2 * D, device=dev, dtype=F32)
    if dgb is None:
        dgb = torch.empty(2, D, device=dev, dtype=F32)
    fn = L.lib().wc_layernorm_bwd_h if dy.dtype == F16 else L.lib().wc_layernorm_bwd      # dy may arrive as fp16 rows
    fn(L.ptr(dy, dy.dtype, "dy"), L.ptr(x, F32, "x"), L.ptr(w, F32, "w"), L.ptr(add, F32, "add"), eps,
       L.ptr(dx32), L.ptr(dx16), float(out_scale), L.ptr(part), L.ptr(dgb), float(alpha), rows, D, L.stream())
    return dx32, dx16, dgb


def layernorm_bwd2(dya, wa, dyb, wb, x, *, add=None, want32=True, want16=False, out_scale=1.0, alpha=1.0, eps=1e-5, dgba=None, dgbb=None):
    """Two LayerNorms of one input x (gammas wa / wb, fp16 gradients dya / dyb) back-propagated in one pass:
    -> (dx32 or None, dx16 or None, dgba, dgbb), dgb* (2, D) = alpha * [dgamma; dbeta] of each norm."""
    rows, D = x.shape
    dev = x.device
    dx32 = torch.empty(rows, D, device=dev, dtype=F32) if want32 else None
    dx16 = torch.empty(rows, D, device=dev, dtype=F16) if want16 else None
    part = torch.empty(min((rows + 15) // 16, 2048) * 4 * D, device=dev, dtype=F32)
    dgba = torch.empty(2, D, device=dev, dtype=F32) if dgba is None else dgba
    dgbb = torch.empty(2, D, device=dev, dtype=F32) if dgbb is None else dgbb
    L.lib().wc_layernorm_bwd2_h(L.ptr(dya, F16, "dya"), L.ptr(wa, F32, "wa"), L.ptr(dyb, F16, "dyb"), L.ptr(wb, F32, "wb"),
                                L.ptr(x, F32, "x"), L.ptr(add, F32, "add"), eps, L.ptr(dx32), L.ptr(dx16), float(out_scale),
                                L.ptr(part), L.ptr(dgba), L.ptr(dgbb), float(alpha), rows, D, L.stream())
    return dx32, dx16, dgba, dgbb


def sigmoid_gram_bwd(dAP, AP, scale=1.0, with_lo=True):
    """-> Split (B, n, np): rows zero padded to np = ceil64(n) so n can be a GEMM K dimension."""
    B, n, _ = AP.shape
    np_ = (n + 63) // 64 * 64
    alloc = torch.zeros if np_ != n else torch.empty
    s = Split(alloc(B, n, np_, device=AP.device, dtype=F16),
              alloc(B, n, np_, device=AP.device, dtype=F16) if with_lo else None)
    L.lib().wc_sigmoid_gram_bwd(L.ptr(dAP, F32, "dAP"), L.ptr(AP, F32, "AP"), L.ptr(s.hi), L.ptr(s.lo), B, n, np_,
                                float(scale), L.stream())
    return s


def colscale_split(x, cs, rows_per_batch, want32=True, with_lo=True, alpha=1.0):
    rows, C = x.shape
    dev = x.device
    out32 = torch.empty(rows, C, device=dev, dtype=F32) if want32 else None
    s = Split(torch.empty(rows, C, device=dev, dtype=F16), torch.empty(rows, C, device=dev, dtype=F16) if with_lo else None)
    L.lib().wc_colscale_split(L.ptr(x, F32, "x"), L.ptr(cs, F32, "cs"), L.ptr(out32), L.ptr(s.hi), L.ptr(s.lo), rows, C,
                              rows_per_batch, float(alpha), L.stream())
    return out32, s


class WeightCache:
    """fp16 MFMA operands (row-major and transposed, hi[+lo]) of a set of fp32 weight matrices, refreshed by ONE
    kernel launch when any of them changed (`Tensor._version`, bumped by the optimizer's in-place update)."""

    def __init__(self):
        self.key = None
        self.versions = None
        self.views = {}

    def refresh(self, named, exact, force=False, lo_names=()):
        """named: list of (name, 2-D contiguous fp32 CUDA tensor) -- or (name, [tensors]) for matrices of equal width that
        are STACKED along the rows into one operand (e.g. two Linears that share their input run as one GEMM).
        force: convert even if no version counter moved (training: an update through `param.data` does not bump
        `param._version`).  lo_names: weights that carry their fp16 remainder in the row-major (forward) copy although
        `exact` is off."""
        L.require_gpu()
        lo_names = frozenset(lo_names)
        named = [(n, list(t) if isinstance(t, (list, tuple)) else [t]) for n, t in named]
        flat = [t for _, ts in named for t in ts]
        key = tuple((n, tuple((t.data_ptr(), tuple(t.shape)) for t in ts)) for n, ts in named) + (bool(exact), lo_names)
        if key != self.key:
            dev = flat[0].device
            tot = sum(t.numel() for t in flat)
            totT = sum(ts[0].shape[1] * ((sum(t.shape[0] for t in ts) + 63) // 64 * 64) for _, ts in named)
            self.hi = torch.empty(tot, device=dev, dtype=F16)
            self.lo = torch.empty(tot, device=dev, dtype=F16) if (exact or lo_names) else None
            self.hiT = torch.zeros(totT, device=dev, dtype=F16)          # K padding of the transposed copies stays 0
            self.loT = torch.zeros(totT, device=dev, dtype=F16) if exact else None
            rows, self.views, o, oT = [], {}, 0, 0
            for n, ts in named:
                R, C = sum(t.shape[0] for t in ts), ts[0].shape[1]
                ldT = (R + 63) // 64 * 64
                has_lo = exact or n in lo_names
                hi, hiT = self.hi[o:o + R * C].view(R, C), self.hiT[oT:oT + C * ldT].view(C, ldT)
                lo = self.lo[o:o + R * C].view(R, C) if has_lo else None
                loT = self.loT[oT:oT + C * ldT].view(C, ldT) if exact else None
                r0 = 0
                for t in ts:            # each source converts into its row block / its column block of the transposed copy
                    if t.shape[1] != C or (len(ts) > 1 and r0 % 4):
                        raise RuntimeError("WeightCache: stacked matrices need equal widths and row counts that are multiples of 4")
                    rows.append([t.data_ptr(), hi.data_ptr() + 2 * r0 * C, (lo.data_ptr() + 2 * r0 * C) if has_lo else 0,
                                 hiT.data_ptr() + 2 * r0, (loT.data_ptr() + 2 * r0) if exact else 0, t.shape[0], C, ldT])
                    r0 += t.shape[0]
                self.views[n] = (Split(hi, lo), Split(hiT, loT), ldT)
                o += R * C
                oT += C * ldT
            self.table = torch.tensor(rows, dtype=torch.int64).pin_memory().to(dev, non_blocking=True)
            self.count = len(rows)
            self.key, self.versions = key, None
        versions = tuple(t._version for t in flat)
        if force or versions != self.versions:
            L.lib().wc_convert_weights(L.ptr(self.table, torch.int64, "table"), self.count, 32, L.stream())
            self.versions = versions

    def w(self, name):
        """Split (R, C): the forward's W operand (K = C contiguous)."""
        return self.views[name][0]

    def wT(self, name):
        """(Split (C, ldT), ldT): the backward's W^T operand (K = R contiguous, zero padded to ldT)."""
        return self.views[name][1], self.views[name][2]
