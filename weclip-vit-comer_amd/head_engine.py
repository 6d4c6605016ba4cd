"""Adapters + decoder + attn_pred on the HIP path, forward AND backward.

reference forward: WeCLIP_model/segformer_head.py:69-80, WeCLIP_model/Decoder/TransDecoder.py:113-125,
WeCLIP_model/model_attn_aff_voc.py:134-137; the reference backward is torch.autograd over those.
Here both directions are explicit launches of the MFMA GEMM (csrc/gemm.hip), LayerNorm
(norm.hip / train_ops.hip), flash attention forward/backward (attention.hip / attention_bwd.hip)
and small reductions; `HeadFunction` exposes the pair to autograd so `loss.backward()` fills
`.grad` of the unchanged nn.Module parameters.

Gradients travel multiplied by GRAD_SCALE (a power of two) because they are MFMA operands in
fp16 (hi[+lo]); parameter gradients are unscaled in the last epilogue.
Weight gradients dW = dY^T X use the same TN GEMM on transposed fp16 copies of dY and X
(tokens become the K dimension, zero padded to a multiple of 64).
"""
import os

import torch

from . import config, ops
from .clip import vit_engine as VE
from .ops import F16, F32, Split

GRAD_SCALE = 4096.0
_WGRAD_WGS = 512   # workgroups a split-K weight-gradient GEMM may use (round 4, same box: 512 -> 12.71, 256 -> 12.90, 128 -> 13.27 ms per step)


def _f(p):
    return p.detach().float().contiguous()


def _uniform_stride(ts):
    """Element stride between consecutive same-shaped tensors of one allocation, or None."""
    p = [t.data_ptr() for t in ts]
    if len(p) < 2:
        return None
    d = p[1] - p[0]
    es = ts[0].element_size()
    if d <= 0 or d % (8 * es) or any(p[i + 1] - p[i] != d for i in range(len(p) - 1)) or \
            any(t.shape != ts[0].shape for t in ts):
        return None
    return d // es


class HeadEngine:
    def __init__(self, fuse, dec):
        self.fuse, self.dec = fuse, dec
        self.E = dec.linear_pred.weight.shape[1]
        self.nc = dec.linear_pred.weight.shape[0]
        self.index = fuse.indexes
        # name -> fp32 buffer the gradient of that parameter is written INTO (not accumulated) by backward();
        # set by TrainStep to views of its flat all-reduce bucket so no per-parameter copy/add kernels run
        self.direct_grads = None
        self.fwd_stream = None          # set by WeCLIP.forward while the head forward runs beside the CAM chain
        self.row_gemm = True            # K = N = 256 Linears on the row-streaming GEMM kernel (False: tile kernels, for A/B)
        self.wcache = ops.WeightCache()     # fp16 (row-major + transposed) copies of every weight matrix, one launch per step

    def params(self):
        return list(self.fuse.parameters()) + list(self.dec.parameters())

    def param_names(self):
        return ["fuse." + n for n, _ in self.fuse.named_parameters()] + \
               ["dec." + n for n, _ in self.dec.named_parameters()]

    def dec_params(self):
        return list(self.dec.parameters())

    def dec_param_names(self):
        return ["dec." + n for n, _ in self.dec.named_parameters()]

    def _weight_matrices(self):
        out = []
        for l, mlp in enumerate(self.fuse.linears_modulelist):
            out.append((f"ad{l}.proj", mlp.proj.weight.detach()))
            out.append((f"ad{l}.proj_2", mlp.proj_2.weight.detach()))
        out.append(("fuse", self.fuse.linear_fuse.weight.detach().flatten(1)))
        out.append(("pred", self.dec.linear_pred.weight.detach().flatten(1)))
        for i, blk in enumerate(self.dec.transformer.resblocks):
            out += [(f"b{i}.in", blk.attn.in_proj_weight.detach()), (f"b{i}.out", blk.attn.out_proj.weight.detach()),
                    (f"b{i}.fc", blk.mlp.c_fc.weight.detach()), (f"b{i}.pj", blk.mlp.c_proj.weight.detach())]
        return [(n, t if t.dtype == F32 and t.is_contiguous() else t.float().contiguous()) for n, t in out]

    # ------------------------------------------------------------------------------ forward
    def forward(self, xs, B, Lq, h, w, drop_scale=None, F_rows=None):
        """xs: `index` Splits of the encoder block outputs (B*L, C) fp16 (CLS row first per image).
        drop_scale (B, E) f32 = Dropout2d mask / (1 - p) or None.  Returns (seg, attn_pred, ctx).
        F_rows (B*h*w, E) f32, if given, replaces the adapters + fuse stage (the ViT-CoMer inserts produce the
        decoder input themselves): only the decoder, linear_pred and attn_pred run here, and backward() returns
        the gradient w.r.t. F_rows under the key "__dF__"."""
        ex = config.exact()
        E, hw, M = self.E, h * w, B * h * w
        n = self.index
        wc = self.wcache
        hl = 63 if ex else int(config.head_lo)
        lo_names = set()
        if hl & 2:
            lo_names.add("fuse")
        for l in range(n):
            if hl & 8:
                lo_names.add(f"ad{l}.proj_2")
            if hl & 16:
                lo_names.add(f"ad{l}.proj")
        wc.refresh(self._weight_matrices(), ex, force=self.fuse.training or self.dec.training,
                   lo_names=() if (ex or F_rows is not None) else lo_names)
        if F_rows is not None:
            dev = F_rows.device
            ctx = dict(B=B, L=Lq, h=h, w=w, xs=None, drop=None, ex=ex, front=False)
            F32_ = F_rows.detach().float().contiguous()
            Fh = ops.split_f16(F32_, True)
            ctx.update(F32=F32_, Fh=Fh)
            return self._decode(ctx, F32_, Fh, B, h, w, ex)
        C = xs[0].hi.shape[1]
        dev = xs[0].hi.device
        ctx = dict(B=B, L=Lq, h=h, w=w, xs=xs, drop=drop_scale, ex=ex, front=True)
        # adapters: t1 = relu(X W1^T + b1); cat[:, l] = t1 W2^T + b2
        cat = Split(torch.empty(M, n * E, device=dev, dtype=F16), torch.empty(M, n * E, device=dev, dtype=F16) if hl & 1 else None)
        t1b = Split(torch.empty(n, M, E, device=dev, dtype=F16), torch.empty(n, M, E, device=dev, dtype=F16) if hl & 4 else None)
        t1s = [Split(t1b.hi[l], t1b.lo[l] if hl & 4 else None) for l in range(n)]
        xlo = bool(hl & 32)
        grp = self._adapter_groups(xs, B, Lq, C, xlo)
        if grp is not None:
            # all n adapters in TWO grouped launches (wc_gemm_f16_grouped) instead of 2 n: the encoder wrote its fp16
            # block outputs into one (n, B*L, C) buffer and the weight cache holds the adapter weights at a uniform stride
            xb, sw1, sw2 = grp
            mods = self.fuse.linears_modulelist
            b1 = torch.stack([m.proj.bias.detach().float() for m in mods])
            b2 = torch.stack([m.proj_2.bias.detach().float() for m in mods])
            a = Split(xb.hi.view(-1)[C:], xb.lo.view(-1)[C:] if (xlo and xb.lo is not None) else None)
            ops.gemm(a, wc.w("ad0.proj"), hw, E, C, bias=b1, out16=t1b.hi, out16lo=t1b.lo, act=2, batch=n * B, zdiv=B,
                     sA=Lq * C, sA2=B * Lq * C, sW=0, sW2=sw1, sC=hw * E, sC2=M * E, sB2=E)
            if self.row_gemm and not ex and cat.lo is None and t1b.lo is None and ops.gemm_row_ok(M, E, E):
                # all n second Linears on the row-streaming kernel (weights stationary in registers, A read once, whole 512-byte
                # output rows into the adapters' column slices of the fuse input): the product is HBM-bound, 185 MB per launch
                ops.gemm_row_grouped(t1b.hi, wc.w("ad0.proj_2").hi, M, E, E, n, bias=b2, out16=cat.hi, ldc16=n * E, gA=M * E, gW=sw2,
                                     gB=E, gC=E)
            else:
                ops.gemm(t1b, wc.w("ad0.proj_2"), M, E, E, bias=b2, out16=cat.hi, out16lo=cat.lo, ldc=n * E, batch=n, zdiv=1,
                         sA2=M * E, sW2=sw2, sC2=E, sB2=E)
        else:
            for l, mlp in enumerate(self.fuse.linears_modulelist):
                a = Split(xs[l].hi.view(-1)[C:], xs[l].lo.view(-1)[C:] if (xlo and xs[l].lo is not None) else None)
                t1 = t1s[l]
                ops.gemm(a, wc.w(f"ad{l}.proj"), hw, E, C, bias=_f(mlp.proj.bias), out16=t1.hi, out16lo=t1.lo,
                         act=2, batch=B, sA=Lq * C, sW=0, sC=hw * E)
                ops.gemm(t1, wc.w(f"ad{l}.proj_2"), M, E, E, bias=_f(mlp.proj_2.bias),
                         out16=cat.hi.view(-1)[l * E:], out16lo=cat.lo.view(-1)[l * E:] if hl & 1 else None, ldc=n * E)
        ctx["t1b"] = t1b
        # fuse (1x1 conv) + Dropout2d
        F32_ = torch.empty(M, E, device=dev, dtype=F32)
        # F always carries its fp16 remainder: the Gram matrix F^T F squares the rounding error of F and sigmoid'(0) = 1/4
        # passes it on (attn_pred abs error 6e-3 with F rounded once, 1e-3 with hi+lo; the GEMM is 8.6 GFLOP)
        Fh = Split(torch.empty(M, E, device=dev, dtype=F16), torch.empty(M, E, device=dev, dtype=F16))
        wf = wc.w("fuse")
        ops.gemm(cat, wf, hw, E, n * E, bias=_f(self.fuse.linear_fuse.bias), out32=F32_, out16=Fh.hi, out16lo=Fh.lo,
                 batch=B, sA=hw * n * E, sW=0, sC=hw * E, cscale=drop_scale, sCS=E)
        ctx.update(cat=cat, t1s=t1s, F32=F32_, Fh=Fh)
        return self._decode(ctx, F32_, Fh, B, h, w, ex)

    def _decode(self, ctx, F32_, Fh, B, h, w, ex):
        """decoder blocks -> linear_pred, and attn_pred = sigmoid(F^T F), from the fused feature rows F."""
        E, hw, M = self.E, h * w, B * h * w
        dev = F32_.device
        wc = self.wcache
        # decoder blocks
        x = F32_
        blocks = []
        for i, blk in enumerate(self.dec.transformer.resblocks):
            pk = VE.BlockPack(blk, exact=ex, pre=dict(in_w=wc.w(f"b{i}.in"), out_w=Split(wc.w(f"b{i}.out").hi, None),
                                                      fc_w=wc.w(f"b{i}.fc"), pj_w=wc.w(f"b{i}.pj")))
            x, bc = self._block_fwd(pk, x, B, hw)
            bc["pk"], bc["blk"], bc["i"] = pk, blk, i
            blocks.append(bc)
        ctx["blocks"] = blocks
        # linear_pred (1x1 conv) on the fp16 copy of the last block output
        x3 = blocks[-1]["x2h"]
        seg_rows = torch.empty(M, self.nc, device=dev, dtype=F32)
        ops.gemm(x3, wc.w("pred"), M, self.nc, E,
                 bias=_f(self.dec.linear_pred.bias), out32=seg_rows)
        seg = seg_rows.view(B, h, w, self.nc).permute(0, 3, 1, 2).contiguous()
        # attn_pred = sigmoid(F^T F) per image
        ap = torch.empty(B, hw, hw, device=dev, dtype=F32)
        ops.gemm(Fh, Fh, hw, hw, E, out32=ap, act=3, batch=B, sA=hw * E, sW=hw * E, sC=hw * hw)
        ctx["ap"] = ap
        return seg, ap, ctx

    def _block_fwd(self, pk, x, B, Lq):
        M, E, H, DH = B * Lq, pk.E, pk.H, pk.DH
        dev = x.device
        ex = pk.exact
        _, a = ops.layernorm(x, pk.ln1_w, pk.ln1_b, with_lo=ex)
        qkv = torch.empty(M, 3 * E, device=dev, dtype=F16)
        ops.gemm(a, pk.in_w, M, 3 * E, E, bias=pk.in_b, out16=qkv, scale=ops.q_scale(DH), scale_cols=E)
        o16, lse, _, o32 = ops.attention(qkv, B, Lq, H, DH, want_mean=False, want_o32=True)
        x1 = torch.empty(M, E, device=dev, dtype=F32)
        ops.gemm(o16, pk.out_w, M, E, E, bias=pk.out_b, resid=x, out32=x1, round16=True)
        _, a2 = ops.layernorm(x1, pk.ln2_w, pk.ln2_b, with_lo=ex)
        z = Split(torch.empty(M, 4 * E, device=dev, dtype=F16), torch.empty(M, 4 * E, device=dev, dtype=F16) if ex else None)
        u32 = torch.empty(M, 4 * E, device=dev, dtype=F32)
        ops.gemm(a2, pk.fc_w, M, 4 * E, E, bias=pk.fc_b, out16=z.hi, out16lo=z.lo, act=1, pre32=u32)
        x2 = torch.empty(M, E, device=dev, dtype=F32)
        x2h = Split(torch.empty(M, E, device=dev, dtype=F16), torch.empty(M, E, device=dev, dtype=F16) if ex else None)
        ops.gemm(z, pk.pj_w, M, E, 4 * E, bias=pk.pj_b, resid=x1, out32=x2, out16=x2h.hi, out16lo=x2h.lo)
        return x2, dict(x=x, a=a, qkv=qkv, o16=o16, o32=o32, lse=lse, x1=x1, a2=a2, u32=u32, z=z, x2h=x2h)

    # ------------------------------------------------------------------------------ backward
    def backward(self, ctx, dseg, dap):
        """dseg (B,nc,h,w) / dap (B,hw,hw) fp32 (either may be None).  Returns {param name: grad}.
        The split-K reductions of all weight gradients are collected and run as ONE launch at the end
        (wc_sum_slices_wb_multi; 16 launches of 5-7 us at the launch floor otherwise)."""
        self._pending = []
        # (the weight-gradient GEMMs on a second stream beside the data-gradient chain were measured neutral in round 3: one stream)
        try:
            grads = self._backward_impl(ctx, dseg, dap)
            self._flush_reductions()
        finally:
            self._pending = None
        return grads

    def _partials(self, *a, **kw):
        return ops.wgrad_partials(*a, **kw)

    def _reduce(self, part, gw, gb, ns, N_, K_, alpha, groups=1, sw=0, sb=0):
        """dW / db = alpha * sum of the `ns` split-K slices of `part`: queued while a backward pass is collecting,
        launched right away otherwise."""
        from . import _lib as L
        if getattr(self, "_pending", None) is None:
            if groups == 1:
                L.lib().wc_sum_slices_wb(L.ptr(part, F32), L.ptr(gw, F32), L.ptr(gb, F32), ns, N_, K_, alpha, L.stream())
            else:
                L.lib().wc_sum_slices_wb_grouped(L.ptr(part, F32), L.ptr(gw, F32), L.ptr(gb, F32), ns, N_, K_, alpha, groups,
                                                 sw, sb, L.stream())
            return
        import struct
        abits = struct.unpack("<I", struct.pack("<f", float(alpha)))[0]
        per = ns * N_ * (K_ + 1)
        for g in range(groups):
            self._pending.append((part, part.data_ptr() + 4 * g * per, gw.data_ptr() + 4 * g * sw, gb.data_ptr() + 4 * g * sb,
                                  ns, N_, K_, abits, 0))

    def _flush_reductions(self):
        jobs = self._pending
        if not jobs:
            return
        import ctypes
        from . import _lib as L
        flat = []
        for j in jobs:
            flat.extend(j[1:])
        arr = (ctypes.c_int64 * len(flat))(*flat)
        L.lib().wc_sum_slices_wb_multi(arr, len(jobs), L.stream())      # the partial buffers stay referenced by `jobs` until here

    def _backward_impl(self, ctx, dseg, dap):
        B, h, w, ex = ctx["B"], ctx["h"], ctx["w"], ctx["ex"]
        E, hw, M, nc, n = self.E, h * w, B * h * w, self.nc, self.index
        dev = ctx["F32"].device
        GS, inv = GRAD_SCALE, 1.0 / GRAD_SCALE
        grads = {}
        wg = lambda dy16, x16, N_, K_, wn, bn, **kw: self._wgrad(dy16, x16, M, N_, K_, inv, grads, wn, bn, **kw)
        # ---- linear_pred
        x3h = ctx["blocks"][-1]["x2h"]
        if dseg is not None:
            wpT, ldT = self.wcache.wT("pred")          # (E, ldT): the nc class columns, zero padded to a multiple of 64
            d = torch.zeros(M, ldT, device=dev, dtype=F32)
            d[:, :nc] = dseg.permute(0, 2, 3, 1).reshape(M, nc)
            _, dS = ops.colscale_split(d, None, M, alpha=GS, want32=False, with_lo=ex)
            dx = torch.empty(M, E, device=dev, dtype=F32)
            ops.gemm(dS, wpT, M, E, ldT, out32=dx)
            wg(dS.hi, x3h.hi, nc, E, "dec.linear_pred.weight", "dec.linear_pred.bias", lda=ldT)
        else:
            dx = torch.zeros(M, E, device=dev, dtype=F32)
            grads["dec.linear_pred.weight"] = torch.zeros(nc, E, 1, 1, device=dev)
            grads["dec.linear_pred.bias"] = torch.zeros(nc, device=dev)
        # ---- decoder blocks, last to first
        for i in reversed(range(len(ctx["blocks"]))):
            dx = self._block_bwd(ctx["blocks"][i], dx, B, hw, f"dec.transformer.resblocks.{i}.", grads, inv)
        # ---- attn_pred = sigmoid(F^T F):  dF += (Z + Z^T) F
        if dap is not None:
            S = ops.sigmoid_gram_bwd(dap.contiguous(), ctx["ap"], scale=GS, with_lo=ex)
            hwp = S.hi.shape[-1]
            FT, Kp = ops.transpose_f16(ctx["F32"], hw, E, batch=B, sSrc=hw * E, oR=hwp)   # (E, B*hwp)
            dF = torch.empty(M, E, device=dev, dtype=F32)
            ops.gemm(S, FT, hw, E, hwp, lda=hwp, ldw=Kp, out32=dF, resid=dx, batch=B, sA=hw * hwp, sW=hwp, sC=hw * E)
        else:
            dF = dx
        if not ctx.get("front", True):          # decoder-only mode: the caller owns everything in front of F
            grads["__dF__"] = dF * inv          # gradients are carried multiplied by GRAD_SCALE
            return grads
        # ---- Dropout2d backward + fuse
        _, dFp = ops.colscale_split(dF, ctx["drop"], hw, want32=False, with_lo=ex)
        cat = ctx["cat"]
        dcat = Split(torch.empty(M, n * E, device=dev, dtype=F16), torch.empty(M, n * E, device=dev, dtype=F16) if ex else None)
        ops.gemm(dFp, self.wcache.wT("fuse")[0], M, n * E, E, out16=dcat.hi, out16lo=dcat.lo)
        wg(dFp.hi, cat.hi, E, n * E, "fuse.linear_fuse.weight", "fuse.linear_fuse.bias")
        # ---- adapters
        xs, Lq = ctx["xs"], ctx["L"]
        C = xs[0].hi.shape[1]
        dt1b = torch.empty(n, M, E, device=dev, dtype=F16)
        swT = _uniform_stride([self.wcache.wT(f"ad{l}.proj_2")[0].hi for l in range(n)])
        grouped = ctx.get("t1b") is not None and swT is not None and os.environ.get("WECLIP_GROUPED_ADAPTERS", "1") != "0"
        if grouped and self.row_gemm and not ex and ops.gemm_row_ok(M, E, E):
            ops.gemm_row_grouped(dcat.hi, self.wcache.wT("ad0.proj_2")[0].hi, M, E, E, n, lda=n * E, out16=dt1b, act=5,
                                 auxh=ctx["t1b"].hi, ldaux=E, gA=E, gW=swT, gC=M * E, gX=M * E)
        elif grouped:       # dt1[l] = (dcat[:, l] W2[l]) * relu'(t1[l]) for all adapters in one grouped launch
            ops.gemm(dcat, self.wcache.wT("ad0.proj_2")[0], M, E, E, lda=n * E, out16=dt1b, act=5, auxh=ctx["t1b"].hi,
                     ldaux=E, batch=n, zdiv=1, sA2=E, sW2=swT, sC2=M * E, sX2=M * E)
        if grouped and self._adapter_wgrads_grouped(ctx, dcat, dt1b, xs, B, Lq, C, M, inv, grads):
            return grads
        for l, mlp in enumerate(self.fuse.linears_modulelist):
            p = f"fuse.linears_modulelist.{l}."
            dt2 = Split(dcat.hi.view(-1)[l * E:], dcat.lo.view(-1)[l * E:] if ex else None)
            t1 = ctx["t1s"][l]
            dt1_16 = dt1b[l]
            if not grouped:
                ops.gemm(dt2, self.wcache.wT(f"ad{l}.proj_2")[0], M, E, E, lda=n * E, out16=dt1_16, act=5, auxh=t1.hi, ldaux=E)
            wg(dt2.hi, t1.hi, E, E, p + "proj_2.weight", p + "proj_2.bias", lda=n * E)
            # X = the hw patch rows of every image of the (B, 1 + hw, C) encoder tokens (CLS rows skipped)
            wg(dt1_16, xs[l].hi, E, C, p + "proj.weight", p + "proj.bias", xmap=(hw, Lq, 1))
        return grads

    def _dest(self, name, shape):
        """Where a parameter gradient is written: the caller-provided buffer (TrainStep's flat gradient
        bucket, `direct_grads`) or a fresh dense tensor."""
        d = self.direct_grads.get(name) if self.direct_grads else None
        if d is not None:
            return d.view(shape)
        return torch.empty(shape, device=self.dec.linear_pred.weight.device, dtype=F32)

    def _adapter_groups(self, xs, B, Lq, C, ex):
        """(stacked block outputs Split (n, B*L, C), element stride between consecutive adapters' `proj` / `proj_2`
        operands in the weight cache) when the adapters can run as grouped launches, else None."""
        n = self.index
        big = getattr(xs, "big", None)
        if os.environ.get("WECLIP_GROUPED_ADAPTERS", "1") == "0" or big is None or len(xs) != n or n < 2 or \
                tuple(big.shape) != (n, B * Lq, C) or (ex and xs.big_lo is None):
            return None
        wc = self.wcache
        st = []
        for key in ("proj", "proj_2"):
            p = [wc.w(f"ad{l}.{key}").hi.data_ptr() for l in range(n)]
            d = p[1] - p[0]
            if d <= 0 or d % 16 or any(p[l + 1] - p[l] != d for l in range(n - 1)) or \
                    any(wc.w(f"ad{l}.{key}").hi.shape != wc.w(f"ad0.{key}").hi.shape for l in range(n)):
                return None
            st.append(d // 2)
        return Split(big, xs.big_lo if ex else None), st[0], st[1]

    def _adapter_wgrads_grouped(self, ctx, dcat, dt1b, xs, B, Lq, C, M, inv, grads):
        """Weight / bias gradients of all n adapters in two grouped weight-gradient GEMMs + two grouped split-K
        reductions (instead of 2 n + 2 n launches), written straight into the caller's gradient bucket.  Needs the
        stacked encoder outputs and bucket views at a uniform stride between adapters; returns False otherwise."""
        n, E, hw = self.index, self.E, ctx["h"] * ctx["w"]
        big = getattr(xs, "big", None)
        if big is None or not self.direct_grads or tuple(big.shape) != (n, B * Lq, C):
            return False
        names = [[f"fuse.linears_modulelist.{l}.{k}" for l in range(n)] for k in ("proj.weight", "proj.bias", "proj_2.weight", "proj_2.bias")]
        dst = [[self.direct_grads.get(nm) for nm in row] for row in names]
        if any(d is None or not d.is_contiguous() for row in dst for d in row):
            return False
        st = [_uniform_stride(row) for row in dst]
        if any(v is None for v in st):
            return False
        from . import _lib as L

        def run(dy, lda, gA, x, ldx, gX, N_, K_, xmap, gw, gb, sw, sb):
            tiles = ops.wgrad_tiles(N_, K_) * n
            ns = max(1, min(_WGRAD_WGS // tiles, M // 256))       # one round of workgroups (2 per CU), few partials
            part, ns = self._partials(dy, x, M, N_, K_, lda=lda, ldx=ldx, slices=ns, bias=True, xmap=xmap, groups=n, gA=gA, gX=gX)
            self._reduce(part, gw, gb, ns, N_, K_, inv, groups=n, sw=sw, sb=sb)

        # proj_2: dY = dcat[:, l*E:(l+1)*E], X = t1[l];   proj: dY = dt1[l], X = patch rows of block output l
        run(dcat.hi, n * E, E, ctx["t1b"].hi, E, M * E, E, E, None, dst[2][0], dst[3][0], st[2], st[3])
        run(dt1b, E, M * E, big, C, B * Lq * C, E, C, (hw, Lq, 1), dst[0][0], dst[1][0], st[0], st[1])
        for row, drow in zip(names, dst):
            for nm, d in zip(row, drow):
                grads[nm] = d
        return True

    def _ln_dest(self, wname, bname, D):
        """(2, D) view over the weight and bias gradient buffers of a LayerNorm when the caller's bucket holds them
        back to back (parameters() order: weight, bias), else None (layernorm_bwd then allocates)."""
        gw = self.direct_grads.get(wname) if self.direct_grads else None
        gb = self.direct_grads.get(bname) if self.direct_grads else None
        if gw is None or gb is None or not gw.is_contiguous() or gb.data_ptr() != gw.data_ptr() + 4 * D:
            return None
        return torch.as_strided(gw, (2, D), (D, 1))

    def _wgrad(self, dy16, x16, M, N_, K_, inv, grads, wname, bname, lda=None, ldx=None, xmap=None):
        """dW (N_, K_) = inv * dY^T X and db (N_) = inv * dY^T 1 from the row-major fp16 operands as they lie
        in memory (csrc/gemm.hip gemm_km_kernel: transposing LDS reads, no operand transposes).  The output has
        few 128x128 tiles and a long contraction (all tokens), so the tokens are split over `ns` slices
        (blockIdx.z); wc_sum_slices_wb sums them straight into the dense weight / bias gradient buffers."""
        tiles = ops.wgrad_tiles(N_, K_)
        ns = 1
        while ns * 2 * tiles <= _WGRAD_WGS and M // (ns * 2) >= 256:
            ns *= 2
        part, ns = self._partials(dy16, x16, M, N_, K_, lda=lda, ldx=ldx, slices=ns, bias=True, xmap=xmap)
        gw, gb = self._dest(wname, (N_, K_)), self._dest(bname, (N_,))
        self._reduce(part, gw, gb, ns, N_, K_, inv)
        grads[wname], grads[bname] = gw, gb

    def _block_bwd(self, c, dx2, B, Lq, prefix, grads, inv):
        pk, blk = c["pk"], c["blk"]
        M, E, H, DH = B * Lq, pk.E, pk.H, pk.DH
        dev = dx2.device
        ex = pk.exact
        wg = lambda dy16, x16, N_, K_, wn, bn, **kw: self._wgrad(dy16, x16, M, N_, K_, inv, grads, wn, bn, **kw)
        # MLP
        _, dx2s = ops.colscale_split(dx2, None, M, want32=False, with_lo=ex)
        du = Split(torch.empty(M, 4 * E, device=dev, dtype=F16), torch.empty(M, 4 * E, device=dev, dtype=F16) if ex else None)
        wT = lambda k: self.wcache.wT(f"b{c['i']}.{k}")[0]
        ops.gemm(dx2s, wT("pj"), M, 4 * E, E, out16=du.hi, out16lo=du.lo, act=4, aux=c["u32"],
                 ldaux=4 * E, rpg=1)
        wg(dx2s.hi, c["z"].hi, E, 4 * E, prefix + "mlp.c_proj.weight", prefix + "mlp.c_proj.bias")
        da2 = torch.empty(M, E, device=dev, dtype=F32)
        ops.gemm(du, wT("fc"), M, E, 4 * E, out32=da2)
        wg(du.hi, c["a2"].hi, 4 * E, E, prefix + "mlp.c_fc.weight", prefix + "mlp.c_fc.bias")
        dx1, g16, dgb2 = ops.layernorm_bwd(da2, c["x1"], pk.ln2_w, add=dx2, want32=True, want16=True, alpha=inv,
                                           dgb=self._ln_dest(prefix + "ln_2.weight", prefix + "ln_2.bias", E))
        grads[prefix + "ln_2.weight"], grads[prefix + "ln_2.bias"] = dgb2[0], dgb2[1]
        # forced-fp16 out-projection (clip/myAtt.py:321): gradient rounded to fp16 on both sides
        do16 = torch.empty(M, E, device=dev, dtype=F16)
        ops.gemm(g16, Split(wT("out").hi, None), M, E, E, out16=do16)
        wg(g16, c["o16"], E, E, prefix + "attn.out_proj.weight", prefix + "attn.out_proj.bias")
        # attention + in-projection
        dqkv = ops.attention_bwd(c["qkv"], do16, c["o32"], c["lse"], B, Lq, H, DH, with_lo=ex)
        da = torch.empty(M, E, device=dev, dtype=F32)
        ops.gemm(dqkv, wT("in"), M, E, 3 * E, out32=da)
        wg(dqkv.hi, c["a"].hi, 3 * E, E, prefix + "attn.in_proj_weight", prefix + "attn.in_proj_bias")
        dx, _, dgb1 = ops.layernorm_bwd(da, c["x"], pk.ln1_w, add=dx1, want32=True, alpha=inv,
                                        dgb=self._ln_dest(prefix + "ln_1.weight", prefix + "ln_1.bias", E))
        grads[prefix + "ln_1.weight"], grads[prefix + "ln_1.bias"] = dgb1[0], dgb1[1]
        return dx




class HeadFunction(torch.autograd.Function):
    """autograd bridge: (params...) -> (seg, attn_pred); backward = HeadEngine.backward."""

    @staticmethod
    def forward(ctx, engine, xs, B, Lq, h, w, drop_scale, *params):
        side = engine.fwd_stream
        if side is not None:
            # forward on a second stream (WeCLIP.forward joins it after the CAM chain); the node itself was created on the
            # caller's stream, so the backward -- and the direct gradient writes the optimizer waits for -- run there
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                seg, ap, c = engine.forward(xs, B, Lq, h, w, drop_scale)
        else:
            seg, ap, c = engine.forward(xs, B, Lq, h, w, drop_scale)
        ctx.engine, ctx.c = engine, c
        return seg, ap

    @staticmethod
    def backward(ctx, dseg, dap):
        eng = ctx.engine
        g = eng.backward(ctx.c, dseg.contiguous() if dseg is not None else None,
                         dap.contiguous() if dap is not None else None)
        ctx.c = None
        direct = eng.direct_grads
        out = []
        for n, p in zip(eng.param_names(), eng.params()):
            d = direct.get(n) if direct else None
            if d is None:
                out.append(g[n].reshape(p.shape))
                continue
            if g[n].data_ptr() != d.data_ptr():      # produced outside _wgrad (LayerNorm, zero heads)
                d.copy_(g[n].reshape(d.shape))
            out.append(None)                          # already in the caller's buffer
        return (None,) * 7 + tuple(out)


class DecoderFunction(torch.autograd.Function):
    """autograd bridge for the decoder-only mode: (F_rows, decoder params...) -> (seg, attn_pred)."""

    @staticmethod
    def forward(ctx, engine, F_rows, B, h, w, *params):
        seg, ap, c = engine.forward(None, B, h * w, h, w, F_rows=F_rows)
        ctx.engine, ctx.c = engine, c
        return seg, ap

    @staticmethod
    def backward(ctx, dseg, dap):
        eng = ctx.engine
        g = eng.backward(ctx.c, dseg.contiguous() if dseg is not None else None,
                         dap.contiguous() if dap is not None else None)
        ctx.c = None
        direct = eng.direct_grads
        out = []
        for n, p in zip(eng.dec_param_names(), eng.dec_params()):
            d = direct.get(n) if direct else None
            if d is None:
                out.append(g[n].reshape(p.shape))
                continue
            if g[n].data_ptr() != d.data_ptr():
                d.copy_(g[n].reshape(d.shape))
            out.append(None)
        return (None, g["__dF__"], None, None, None) + tuple(out)
