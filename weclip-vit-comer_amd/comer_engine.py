"""The ViT-CoMer inserts as ONE explicit forward / backward engine on the HIP path.

There is no CoMer code in the reference repository (only ViT_CoMer.pdf §3.2-3.3 and the brief): the network is this
package's own `WeCLIP_model.comer.CoMerInteraction` (MRFP + bidirectional CTI with multi-scale deformable attention on the
adapter outputs of four ViT blocks, fused by a 1x1 conv into the decoder input).  Its module-by-module autograd form
(comer.py + hip_functional.py) launches ~1 070 kernels per pass, a third of them conversions, transposes, concatenations
and element-wise glue; this engine runs the same arithmetic on token rows end to end:

  * every activation that feeds a GEMM leaves its producer as the fp16 MFMA operand (LayerNorm, GEMM epilogue, deformable
    attention, the depth-wise conv kernel): no split / cast passes; all weights are converted (row-major and transposed)
    by ONE launch per step (ops.WeightCache), `sampling_offsets` and `attention_weights` run as one GEMM on stacked weights;
  * residual adds, the CTI gate `gamma`, the exact GELU (forward and derivative) live in GEMM epilogues (csrc/gemm.hip
    act 6 / 7, cscale, resid); the soft-max over the sampling weights and the sampling-location arithmetic are one small
    kernel each way (csrc/comer.hip msda_prep_*);
  * MRFP's depth-wise 3x3 / 5x5 convolutions of all pyramid levels are one launch on NHWC rows with the GELU fused
    (csrc/comer.hip mrfp_dwconv_*): no NCHW transposes, no channel concatenations;
  * the backward pass is written out (no autograd graph inside): weight gradients by the split-K row-major GEMM
    (gemm_km), their reductions collected into one launch, gradients carried multiplied by GRAD_SCALE like the head's.

The conv stem (`SpatialPrior`) stays on its autograd Functions (hip_functional.conv3x3_rows / groupnorm_relu_rows); the
engine starts from its output and hands its gradient back.  With `adapters` set (`CoMerInteraction.forward_tokens`) the four
WeCLIP adapter MLPs of the stage blocks run inside the engine on the encoder's f16 block outputs, and with
`CoMerInteraction.direct_grads` (set by train_step.TrainStep) every parameter gradient is written straight into the views of
the flat all-reduce bucket.  Parity: against an fp64 CPU evaluation of the same network and the module-by-module form
(tests/test_comer_gpu.py); no reference code exists ("parity unpinned").  Runs in `fast` precision (single fp16 operands).
"""
import ctypes

import torch

from . import _lib as L
from . import config, ops
from .ops import F16, F32, Split

GS = 4096.0
INV = 1.0 / GS


# workgroups a split-K weight-gradient GEMM may use: the inserts reduce ~50 partial sets per step, so fewer, longer slices pay
# (inserts alone, same box: 1024 -> 12.99, 512 -> 12.91, 256 -> 12.82, 128 -> 12.44 vs 11.92 ms at 256 after the load fixes)
_WGRAD_WGS = 256


def _shape_array(shapes):
    return L.int_array([v for hw in shapes for v in hw])


class ComerEngine:
    def __init__(self, net):
        self.net = net
        self.wc = ops.WeightCache()
        self._ref = {}
        self.adapters = None        # the 4 WeCLIP adapter MLPs (segformer_head.MLP) when the engine also runs them (forward_tokens)
        self.row_gemm = True        # Linear layers on the row-streaming GEMM kernel (False: the tile kernels + LayerNorm launches, for A/B)

    # ------------------------------------------------------------------------------------------ parameters
    def params(self):
        """Every parameter the engine differentiates (all of the inserts except the conv stem), in a fixed order."""
        net = self.net
        out = []
        for m, c in zip(net.mrfp, net.cti):
            out += [m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias, m.dw3.weight, m.dw3.bias, m.dw5.weight, m.dw5.bias]
            out += [c.nv_q.weight, c.nv_q.bias, c.nv_f.weight, c.nv_f.bias, c.gamma, c.nc_q.weight, c.nc_q.bias,
                    c.nc_f.weight, c.nc_f.bias, c.ffn_norm.weight, c.ffn_norm.bias,
                    c.ffn[0].weight, c.ffn[0].bias, c.ffn[2].weight, c.ffn[2].bias]
            for a in (c.to_v, c.to_c):
                out += [a.sampling_offsets.weight, a.sampling_offsets.bias, a.attention_weights.weight, a.attention_weights.bias,
                        a.value_proj.weight, a.value_proj.bias, a.output_proj.weight, a.output_proj.bias]
        out += [net.fuse.weight, net.fuse.bias]
        for a in (self.adapters or []):
            out += [a.proj.weight, a.proj.bias, a.proj_2.weight, a.proj_2.bias]
        return out

    def _mats(self):
        f = lambda t: t.detach() if (t.dtype == F32 and t.is_contiguous()) else t.detach().float().contiguous()
        out = []
        for i, (m, c) in enumerate(zip(self.net.mrfp, self.net.cti)):
            out += [(f"m{i}.fc1", f(m.fc1.weight)), (f"m{i}.fc2", f(m.fc2.weight)), (f"f{i}.0", f(c.ffn[0].weight)),
                    (f"f{i}.2", f(c.ffn[2].weight))]
            for tag, a in (("v", c.to_v), ("c", c.to_c)):
                out += [(f"{tag}{i}.vp", f(a.value_proj.weight)), (f"{tag}{i}.op", f(a.output_proj.weight)),
                        (f"{tag}{i}.ow", [f(a.sampling_offsets.weight), f(a.attention_weights.weight)])]
        out.append(("fuse", f(self.net.fuse.weight).flatten(1)))
        for i, a in enumerate(self.adapters or []):
            out += [(f"a{i}.p1", f(a.proj.weight)), (f"a{i}.p2", f(a.proj_2.weight))]
        return out

    def ref_points(self, shapes, dev):
        """(sum H*W, 2) reference points (x, y) in [0,1] of the pixels of the given maps, in row order."""
        key = (tuple(shapes), str(dev))
        if key not in self._ref:
            pts = []
            for h, w in shapes:
                ys, xs = torch.meshgrid((torch.arange(h, device=dev) + 0.5) / h, (torch.arange(w, device=dev) + 0.5) / w, indexing="ij")
                pts.append(torch.stack([xs.reshape(-1), ys.reshape(-1)], -1))
            self._ref[key] = torch.cat(pts, 0).float().contiguous()
        return self._ref[key]

    # ------------------------------------------------------------------------------------------ small wrappers
    @staticmethod
    def _b(p):
        return p.detach().float().contiguous()

    def _ln16(self, x, ln):
        return ops.layernorm(x, self._b(ln.weight), self._b(ln.bias), eps=ln.eps, want32=False, want16=True)[1].hi

    def _msda_fwd(self, value, shapes, ow, ld, att, ref, B, Lq):
        """value (B*S, 256) f16, ow (B*Lq, ld) f32 -> (o16 (B*Lq, 256) f16, loc, attn)."""
        M, P, nL = att.n_heads, att.n_points, len(shapes)
        D = att.d_model // M
        dev = value.device
        loc = torch.empty(B * Lq * M * nL * P * 2, device=dev, dtype=F32)
        attn = torch.empty(B * Lq * M * nL * P, device=dev, dtype=F32)
        hs = _shape_array(shapes)
        lib = L.lib()
        o16 = torch.empty(B * Lq, M * D, device=dev, dtype=F16)
        if lib.cdll.wc_msda_fused_supported(nL, M, D, P):        # locations + soft-max inside the attention kernel
            lib.wc_msda_fwd_f(L.ptr(value, F16), 1, hs, nL, L.ptr(ow, F32), ld, L.ptr(self._b(att.sampling_offsets.bias), F32),
                              L.ptr(self._b(att.attention_weights.bias), F32), L.ptr(ref, F32), 1, L.ptr(loc), L.ptr(attn), None,
                              L.ptr(o16), B, Lq, M, D, P, L.stream())
            return o16, loc, attn
        lib.wc_msda_prep_fwd(L.ptr(ow, F32), L.ptr(self._b(att.sampling_offsets.bias), F32), L.ptr(self._b(att.attention_weights.bias), F32),
                             L.ptr(ref, F32), L.ptr(loc), L.ptr(attn), hs, nL, B, Lq, M, P, ld, 1, L.stream())
        lib.wc_msda_fwd_h(L.ptr(value, F16), 1, hs, nL, L.ptr(loc), L.ptr(attn), None, L.ptr(o16), B, Lq, M, D, P, L.stream())
        return o16, loc, attn

    def _msda_bwd(self, value, shapes, loc, attn, gout, att, B, Lq, ld):
        """-> (dvalue16 (B*S, 256) f16, dow16 (B*Lq, ld) f16 with zeroed padding columns)."""
        M, P, nL = att.n_heads, att.n_points, len(shapes)
        D = att.d_model // M
        S = sum(h * w for h, w in shapes)
        dev = value.device
        gv = torch.empty(value.shape, device=dev, dtype=F16)          # (value is f16 already)
        gmax = torch.empty(1, device=dev, dtype=torch.int32)
        ws = torch.empty(B * M * (2 * S + nL * Lq * P * 8), device=dev, dtype=torch.int32)
        hs = _shape_array(shapes)
        lib = L.lib()
        dow16 = torch.empty(B * Lq, ld, device=dev, dtype=F16)
        if lib.cdll.wc_msda_fused_supported(nL, M, D, P):        # soft-max / location backward inside the attention kernel
            lib.wc_msda_bwd_f(L.ptr(value, F16), 1, hs, nL, L.ptr(loc), L.ptr(attn), L.ptr(gout, F16, "gout"), 1, None, L.ptr(gv),
                              L.ptr(dow16), ld, L.ptr(gmax), L.ptr(ws), B, Lq, M, D, P, L.stream())
            return gv, dow16
        gl, ga = torch.empty_like(loc), torch.empty_like(attn)
        lib.wc_msda_bwd_h(L.ptr(value, F16), 1, hs, nL, L.ptr(loc), L.ptr(attn), L.ptr(gout, F16, "gout"), 1, None, L.ptr(gv), L.ptr(gl),
                          L.ptr(ga), L.ptr(gmax), L.ptr(ws), B, Lq, M, D, P, L.stream())
        lib.wc_msda_prep_bwd(L.ptr(gl), L.ptr(ga), L.ptr(attn), None, L.ptr(dow16), hs, nL, B, Lq, M, P, ld, L.stream())
        return gv, dow16

    def _mm(self, a, w, M, N, K, *, ln=(), cscale=None, ldc16=None, **kw):
        """GEMM dispatch: the row-streaming kernel (csrc/gemm_row.hip: weights stationary, whole-row epilogue, LayerNorm of the
        output fused) where the shape allows, else the tile kernels + separate LayerNorm launches.  `ln`: [(LayerNorm module,
        fp16 out (M, N))] of the finished fp32 row; `cscale`: (N,) column scale after the bias."""
        a16 = a.hi if isinstance(a, Split) else a
        w16 = w.hi if isinstance(w, Split) else w
        # (the row kernel streams at the HBM rate -- +8 192 rows per 1.7 us -- behind a fixed ~10 us (every workgroup first loads the
        #  whole weight matrix into its registers): at 16 384 rows the tile kernels win unless a LayerNorm launch is saved)
        big = M >= 32768 or bool(ln) or ldc16 is not None
        if self.row_gemm and big and ops.gemm_row_ok(M, N, K) and (not ln or N == 256) and kw.get("lda", K) % 8 == 0:
            kw.pop("rpg", None)
            ops.gemm_row(a16, w16, M, N, K, cscale=cscale, ldc16=ldc16, eps=ln[0][0].eps if ln else 1e-5,
                         ln=[(self._b(m.weight), self._b(m.bias), o) for m, o in ln], **kw)
            return
        if ldc16 is not None and kw.get("out16") is not None:
            raise RuntimeError("ComerEngine._mm: a separate fp16 pitch needs the row kernel")
        if cscale is not None:
            kw.update(cscale=cscale.view(1, N), sCS=0)
        ops.gemm(a16, w16, M, N, K, **kw)
        for m, o in ln:
            L.lib().wc_layernorm(L.ptr(kw["out32"], F32, "x"), N, L.ptr(self._b(m.weight), F32), L.ptr(self._b(m.bias), F32), m.eps, None,
                                 L.ptr(o, F16, "ln.out"), None, M, N, L.stream())

    @staticmethod
    def _f16(x, alpha=1.0, cs=None):
        """fp32 rows -> fp16 operand (optionally x alpha, x a column scale)."""
        rows = x.shape[0]
        return ops.colscale_split(x, cs, rows, want32=False, with_lo=False, alpha=alpha)[1].hi

    # ------------------------------------------------------------------------------------------ forward
    def forward(self, c0, shapes, hw, maps, x16=None, Lq=None):
        """c0 (B, S, C) pyramid tokens of the stem; shapes [(H, W)] x 3; maps: the 4 adapter outputs (B, h*w, C) -- or, with
        `self.adapters` set, x16: the 4 encoder block outputs (B*Lq, Cin) f16 (CLS row first per image, straight from the
        encoder's epilogue) from which the engine computes the adapter outputs proj_2(relu(proj(tokens))) itself
        (WeCLIP_model/segformer_head.py:13-28).  -> (y rows (B*h*w, C) f32, ctx)."""
        net = self.net
        B, S, C = c0.shape
        h, w = hw
        nhw = h * w
        Mc, Mv = B * S, B * nhw
        dev = c0.device
        lib = L.lib()
        self.wc.refresh(self._mats(), False, force=True)
        W = self.wc.w
        hs3 = _shape_array(shapes)
        n16 = shapes[0][0] * shapes[0][1]              # the 1/16 level's rows inside an image's S rows
        if shapes[1] != (h, w):
            raise RuntimeError("ComerEngine: the stem's middle level must be the 1/16 map")
        ref_v = self.ref_points([(h, w)], dev)
        ref_c = self.ref_points(shapes, dev)
        c = c0.detach().float().contiguous().view(Mc, C)
        c16 = torch.empty(Mc, C, device=dev, dtype=F16)
        lib.wc_rows_copy_f16(L.ptr(c, F32), 1, L.ptr(c16), 1, Mc, C, C, 0, C, 0, L.stream())
        nst = len(net.stage_blocks)
        cat16 = torch.empty(Mv, 2 * nst * C, device=dev, dtype=F16)
        st = []
        for i in range(nst):
            m, t = net.mrfp[i], net.cti[i]
            hid = m.fc1.weight.shape[0]
            s = {}
            e16 = lambda rows: torch.empty(rows, C, device=dev, dtype=F16)
            q1 = e16(Mv)
            if x16 is not None:        # the WeCLIP adapter of this stage's ViT block, on the patch rows of the f16 tokens
                ad = self.adapters[i]
                Cin = x16[i].shape[1]
                t1 = torch.empty(Mv, C, device=dev, dtype=F16)
                ops.gemm(x16[i].view(-1)[Cin:], W(f"a{i}.p1"), nhw, C, Cin, bias=self._b(ad.proj.bias), out16=t1, act=2, batch=B,
                         sA=Lq * Cin, sW=0, sC=nhw * C)
                v = torch.empty(Mv, C, device=dev, dtype=F32)
                self._mm(t1, W(f"a{i}.p2"), Mv, C, C, bias=self._b(ad.proj_2.bias), out32=v, ln=[(t.nv_q, q1)])      # q1 = LN(v)
                s.update(t1=t1, x16=x16[i], Lq=Lq)
            else:
                v = maps[i].detach().float().contiguous().view(Mv, C)
                q1 = self._ln16(v, t.nv_q)
            # ---- MRFP: c1 = c + fc2(gelu(dwconv(fc1(c))))
            x1 = torch.empty(Mc, hid, device=dev, dtype=F32)
            self._mm(c16, W(f"m{i}.fc1"), Mc, hid, C, bias=self._b(m.fc1.bias), out32=x1)
            x2 = torch.empty(Mc, hid, device=dev, dtype=F32)
            g16 = torch.empty(Mc, hid, device=dev, dtype=F16)
            lib.wc_mrfp_dwconv_fwd(L.ptr(x1), L.ptr(self._b(m.dw3.weight).view(-1), F32), L.ptr(self._b(m.dw3.bias), F32),
                                   L.ptr(self._b(m.dw5.weight).view(-1), F32), L.ptr(self._b(m.dw5.bias), F32), L.ptr(x2), L.ptr(g16),
                                   hs3, len(shapes), B, hid, L.stream())
            c1 = torch.empty(Mc, C, device=dev, dtype=F32)
            f1, q2 = e16(Mc), e16(Mc)         # LN(c1) with the parameters of nv_f (values of CTI-toV) and of nc_q (queries of CTI-toC)
            self._mm(g16, W(f"m{i}.fc2"), Mc, C, hid, bias=self._b(m.fc2.bias), resid=c, out32=c1, ln=[(t.nv_f, f1), (t.nc_q, q2)])
            # ---- CTI-toV: v1 = v + gamma * out_proj(msda(LN(v) -> offsets / weights, value_proj(LN(c1))))
            val1 = torch.empty(Mc, C, device=dev, dtype=F16)          # fp16 values: half the gather traffic of the deformable attention
            self._mm(f1, W(f"v{i}.vp"), Mc, C, C, bias=self._b(t.to_v.value_proj.bias), out16=val1)
            n1 = W(f"v{i}.ow").hi.shape[0]
            ld1 = (n1 + 63) // 64 * 64
            ow1 = torch.empty(Mv, ld1, device=dev, dtype=F32)
            self._mm(q1, W(f"v{i}.ow"), Mv, n1, C, out32=ow1, ldc=ld1)
            o1, loc1, at1 = self._msda_fwd(val1, shapes, ow1, ld1, t.to_v, ref_v, B, nhw)
            v1 = torch.empty(Mv, C, device=dev, dtype=F32)
            f2 = e16(Mv)
            row = self.row_gemm       # (the row kernel also drops v1's fp16 copy into its column slice of the fuse input)
            self._mm(o1, W(f"v{i}.op"), Mv, C, C, bias=self._b(t.to_v.output_proj.bias), cscale=self._b(t.gamma), resid=v, out32=v1,
                     out16=cat16.view(-1)[2 * i * C:] if row else None, ldc16=2 * nst * C if row else None, ln=[(t.nc_f, f2)])
            if not row:
                lib.wc_rows_copy_f16(L.ptr(v1, F32), 1, L.ptr(cat16.view(-1)[2 * i * C:]), 1, Mv, C, C, 0, 2 * nst * C, 0, L.stream())
            # ---- CTI-toC: c2 = c1 + out_proj(msda(LN(c1), value_proj(LN(v1))))
            val2 = torch.empty(Mv, C, device=dev, dtype=F16)
            self._mm(f2, W(f"c{i}.vp"), Mv, C, C, bias=self._b(t.to_c.value_proj.bias), out16=val2)
            n2 = W(f"c{i}.ow").hi.shape[0]
            ld2 = (n2 + 63) // 64 * 64
            ow2 = torch.empty(Mc, ld2, device=dev, dtype=F32)
            self._mm(q2, W(f"c{i}.ow"), Mc, n2, C, out32=ow2, ldc=ld2)
            o2, loc2, at2 = self._msda_fwd(val2, [(h, w)], ow2, ld2, t.to_c, ref_c, B, S)
            c2 = torch.empty(Mc, C, device=dev, dtype=F32)
            n3 = e16(Mc)
            self._mm(o2, W(f"c{i}.op"), Mc, C, C, bias=self._b(t.to_c.output_proj.bias), resid=c1, out32=c2, ln=[(t.ffn_norm, n3)])
            # ---- FFN: c3 = c2 + ffn2(gelu(ffn0(LN(c2))))
            u = torch.empty(Mc, C, device=dev, dtype=F32)
            g2 = torch.empty(Mc, C, device=dev, dtype=F16)
            self._mm(n3, W(f"f{i}.0"), Mc, C, C, bias=self._b(t.ffn[0].bias), pre32=u, out16=g2, act=6)
            c3 = torch.empty(Mc, C, device=dev, dtype=F32)
            c3_16 = torch.empty(Mc, C, device=dev, dtype=F16)
            self._mm(g2, W(f"f{i}.2"), Mc, C, C, bias=self._b(t.ffn[2].bias), resid=c2, out32=c3, out16=c3_16)
            lib.wc_rows_copy_f16(L.ptr(c3_16.view(-1)[n16 * C:], F16), 0, L.ptr(cat16.view(-1)[(2 * i + 1) * C:]), B, nhw, C, C, S * C,
                                 2 * nst * C, nhw * 2 * nst * C, L.stream())
            s.update(c16=c16, x1=x1, x2=x2, g16=g16, c1=c1, v=v, q1=q1, f1=f1, val1=val1, loc1=loc1, at1=at1, o1=o1, v1=v1, ld1=ld1,
                     n1=n1, q2=q2, f2=f2, val2=val2, loc2=loc2, at2=at2, o2=o2, c2=c2, ld2=ld2, n2=n2, n3=n3, u=u, g2=g2)
            st.append(s)
            c, c16 = c3, c3_16
        y = torch.empty(Mv, C, device=dev, dtype=F32)
        ops.gemm(cat16, W("fuse"), Mv, C, 2 * nst * C, bias=self._b(net.fuse.bias), out32=y)
        ctx = dict(st=st, cat16=cat16, B=B, S=S, C=C, hw=(h, w), shapes=[tuple(x) for x in shapes], n16=n16)
        return y, ctx

    # ------------------------------------------------------------------------------------------ backward
    def _direct(self, p):
        """Where a parameter's gradient may be WRITTEN (not accumulated): its `.grad` when the owner opted in
        (`CoMerInteraction.direct_grads`, set by TrainStep: the flat all-reduce bucket is zeroed every step and every gradient
        is produced exactly once per backward), else None -> a fresh tensor handed back to autograd."""
        rng = getattr(self.net, "direct_grads", False)
        if p is None or not rng:
            return None
        g = p.grad
        if g is None or not g.is_contiguous() or g.dtype != F32 or g.data_ptr() % 16:
            return None
        # only views of THE bucket that opted in are overwritten: any other .grad (gradient accumulation, a second TrainStep
        # without a bucket, a harness calling backward twice) gets a fresh tensor that autograd accumulates as usual
        if rng is not True and not (rng[0] <= g.data_ptr() < rng[1]):
            return None
        return g

    def _wgrad(self, dy16, x16, M, N, K, grads, pw, pb, lda=None, xmap=None):
        """pw.grad / pb.grad (unscaled) of y = x W^T + b from the fp16 operands dy16 (M, lda >= N) [x GS] and x16 (M, K)."""
        tiles = ops.wgrad_tiles(N, K)
        ns = 1
        while ns * 2 * tiles <= _WGRAD_WGS and M // (ns * 2) >= 256:
            ns *= 2
        part, ns = ops.wgrad_partials(dy16, x16, M, N, K, lda=lda, slices=ns, bias=True, xmap=xmap)
        gw, gb = self._direct(pw), self._direct(pb)
        dw = gw.view(N, K) if gw is not None else torch.empty(N, K, device=dy16.device, dtype=F32)
        db = gb if gb is not None else torch.empty(N, device=dy16.device, dtype=F32)
        self._pending.append((part, dw, db, ns, N, K, 0, N))
        if pw is not None:
            grads[id(pw)] = dw.view(pw.shape)
        if pb is not None:
            grads[id(pb)] = db
        return dw, db

    def _flush(self):
        import struct
        jobs = self._pending
        if not jobs:
            return
        abits = struct.unpack("<I", struct.pack("<f", INV))[0]
        flat = []
        for part, dw, db, ns, N, K, r0, Ntot in jobs:          # rows [r0, r0 + N) of a (ns, Ntot, K + 1) partial matrix
            flat += [part.data_ptr() + 4 * r0 * (K + 1), dw.data_ptr(), db.data_ptr(), ns, N, K, abits, Ntot * (K + 1)]
        arr = (ctypes.c_int64 * len(flat))(*flat)
        L.lib().wc_sum_slices_wb_multi(arr, len(jobs), L.stream())

    def backward(self, ctx, dy):
        """dy (B*h*w, C) f32 -> (dc0 (B, S, C), [dv x 4], {id(param): grad})."""
        self._pending = []
        try:
            out = self._backward(ctx, dy)
            self._flush()
        finally:
            self._pending = None
        return out

    def _ln_dest(self, ln):
        """(2, D) destination of a LayerNorm's [dgamma; dbeta]: the adjacent views of the gradient bucket, or None."""
        gw, gb = self._direct(ln.weight), self._direct(ln.bias)
        if gw is not None and gb is not None and gb.data_ptr() == gw.data_ptr() + 4 * gw.numel():
            return torch.as_strided(gw, (2, gw.numel()), (gw.numel(), 1))
        return None

    def _ln_bwd2(self, dya, lna, dyb, lnb, x, add, grads):
        """Both LayerNorms of one input in one pass (csrc/train_ops.hip ln_bwd_kernel<.., true>): -> (dx f32, dx f16)."""
        dx, dx16, ga, gb = ops.layernorm_bwd2(dya, self._b(lna.weight), dyb, self._b(lnb.weight), x, add=add, want32=True,
                                              want16=True, alpha=INV, eps=lna.eps, dgba=self._ln_dest(lna), dgbb=self._ln_dest(lnb))
        grads[id(lna.weight)], grads[id(lna.bias)] = ga[0], ga[1]
        grads[id(lnb.weight)], grads[id(lnb.bias)] = gb[0], gb[1]
        return dx, dx16

    def _ln_bwd(self, dy, x, ln, add, grads, want16=False):
        """dx = LN_bwd(dy) + add (f32) [and its f16 copy: the next GEMMs' operand, no separate conversion pass]."""
        gw, gb = self._direct(ln.weight), self._direct(ln.bias)
        dest = None
        if gw is not None and gb is not None and gb.data_ptr() == gw.data_ptr() + 4 * gw.numel():
            dest = torch.as_strided(gw, (2, gw.numel()), (gw.numel(), 1))       # weight / bias gradients adjacent in the bucket
        dx, dx16, dgb = ops.layernorm_bwd(dy, x, self._b(ln.weight), add=add, want32=True, want16=want16, alpha=INV, eps=ln.eps,
                                          dgb=dest)
        grads[id(ln.weight)], grads[id(ln.bias)] = dgb[0], dgb[1]
        return (dx, dx16) if want16 else dx

    def _backward(self, ctx, dy):
        net = self.net
        B, S, C = ctx["B"], ctx["S"], ctx["C"]
        h, w = ctx["hw"]
        shapes, n16 = ctx["shapes"], ctx["n16"]
        nhw = h * w
        Mc, Mv = B * S, B * nhw
        dev = dy.device
        lib = L.lib()
        WT = lambda n: self.wc.wT(n)[0]
        nst = len(net.stage_blocks)
        grads, dvs = {}, [None] * nst
        hs3 = _shape_array(shapes)
        # ---- fuse conv: dcat pieces and the fuse weight gradient (everything from here on is multiplied by GS)
        dy16 = self._f16(dy.float().contiguous(), alpha=GS)
        self._wgrad(dy16, ctx["cat16"], Mv, C, 2 * nst * C, grads, net.fuse.weight, net.fuse.bias)
        fuseT = WT("fuse").hi                                   # (2*nst*C, ldT = C)
        pieces = []
        for j in range(2 * nst):
            d = torch.empty(Mv, C, device=dev, dtype=F32)
            self._mm(dy16, fuseT[j * C:(j + 1) * C], Mv, C, C, out32=d)
            pieces.append(d)
        dc3 = None
        for i in reversed(range(nst)):
            m, t, s = net.mrfp[i], net.cti[i], ctx["st"][i]
            hid = m.fc1.weight.shape[0]
            if dc3 is None:
                dc3 = torch.zeros(Mc, C, device=dev, dtype=F32)
            lib.wc_rows_add_f32(L.ptr(pieces[2 * i + 1], F32), L.ptr(dc3.view(-1)[n16 * C:], F32), B, nhw, C, C, nhw * C, S * C, 1.0,
                                L.stream())
            # ---- FFN
            dc3_16 = self._f16(dc3)
            du16 = torch.empty(Mc, C, device=dev, dtype=F16)
            self._mm(dc3_16, WT(f"f{i}.2"), Mc, C, C, out16=du16, act=7, aux=s["u"], ldaux=C, rpg=1)
            self._wgrad(dc3_16, s["g2"], Mc, C, C, grads, t.ffn[2].weight, t.ffn[2].bias)
            dn3 = torch.empty(Mc, C, device=dev, dtype=F16)            # (gradients that only feed a LayerNorm backward travel as fp16)
            self._mm(du16, WT(f"f{i}.0"), Mc, C, C, out16=dn3)
            self._wgrad(du16, s["n3"], Mc, C, C, grads, t.ffn[0].weight, t.ffn[0].bias)
            dc2, dc2_16 = self._ln_bwd(dn3, s["c2"], t.ffn_norm, dc3, grads, want16=True)
            # ---- CTI-toC
            do2 = torch.empty(Mc, C, device=dev, dtype=F16)
            self._mm(dc2_16, WT(f"c{i}.op"), Mc, C, C, out16=do2)
            self._wgrad(dc2_16, s["o2"], Mc, C, C, grads, t.to_c.output_proj.weight, t.to_c.output_proj.bias)
            dval2_16, dow2 = self._msda_bwd(s["val2"], [(h, w)], s["loc2"], s["at2"], do2, t.to_c, B, S, s["ld2"])
            dq2 = torch.empty(Mc, C, device=dev, dtype=F16)
            self._mm(dow2, WT(f"c{i}.ow"), Mc, C, s["ld2"], out16=dq2)
            self._ow_grads(dow2, s["q2"], Mc, s["n2"], s["ld2"], t.to_c, grads)
            # (LN_nc_q(c1) and LN_nv_f(c1) normalise the same rows: their backward runs as ONE pass below, once df1 exists)
            df2 = torch.empty(Mv, C, device=dev, dtype=F16)
            self._mm(dval2_16, WT(f"c{i}.vp"), Mv, C, C, out16=df2)
            self._wgrad(dval2_16, s["f2"], Mv, C, C, grads, t.to_c.value_proj.weight, t.to_c.value_proj.bias)
            dv1, dv1_16 = self._ln_bwd(df2, s["v1"], t.nc_f, pieces[2 * i], grads, want16=True)
            # ---- CTI-toV: v1 = v + gamma * (o1 Wop^T + bop)
            gam = self._b(t.gamma)
            gdv1_16 = self._f16(dv1, cs=gam.view(1, C))
            do1 = torch.empty(Mv, C, device=dev, dtype=F16)
            self._mm(gdv1_16, WT(f"v{i}.op"), Mv, C, C, out16=do1)
            G, gsum = self._wgrad(dv1_16, s["o1"], Mv, C, C, grads, None, None)          # G = dv1^T o1, gsum = dv1^T 1 (unscaled)
            self._gamma_jobs.append((t, G, gsum))
            dval1_16, dow1 = self._msda_bwd(s["val1"], shapes, s["loc1"], s["at1"], do1, t.to_v, B, nhw, s["ld1"])
            dq1 = torch.empty(Mv, C, device=dev, dtype=F16)
            self._mm(dow1, WT(f"v{i}.ow"), Mv, C, s["ld1"], out16=dq1)
            self._ow_grads(dow1, s["q1"], Mv, s["n1"], s["ld1"], t.to_v, grads)
            if "t1" in s:              # the adapter MLP behind v: v = t1 W2^T + b2, t1 = relu(x W1^T + b1); x is frozen
                ad = self.adapters[i]
                dvs[i], dv16 = self._ln_bwd(dq1, s["v"], t.nv_q, dv1, grads, want16=True)
                self._wgrad(dv16, s["t1"], Mv, C, C, grads, ad.proj_2.weight, ad.proj_2.bias)
                dt1 = torch.empty(Mv, C, device=dev, dtype=F16)
                self._mm(dv16, WT(f"a{i}.p2"), Mv, C, C, out16=dt1, act=5, auxh=s["t1"], ldaux=C)
                self._wgrad(dt1, s["x16"], Mv, C, s["x16"].shape[1], grads, ad.proj.weight, ad.proj.bias, xmap=(nhw, s["Lq"], 1))
                dvs[i] = None
            else:
                dvs[i] = self._ln_bwd(dq1, s["v"], t.nv_q, dv1, grads)
            df1 = torch.empty(Mc, C, device=dev, dtype=F16)
            self._mm(dval1_16, WT(f"v{i}.vp"), Mc, C, C, out16=df1)
            self._wgrad(dval1_16, s["f1"], Mc, C, C, grads, t.to_v.value_proj.weight, t.to_v.value_proj.bias)
            if t.nc_q.eps == t.nv_f.eps and C <= 256:
                dc1, dc1_16 = self._ln_bwd2(dq2, t.nc_q, df1, t.nv_f, s["c1"], dc2, grads)
            else:
                dc1 = self._ln_bwd(dq2, s["c1"], t.nc_q, dc2, grads)
                dc1, dc1_16 = self._ln_bwd(df1, s["c1"], t.nv_f, dc1, grads, want16=True)
            # ---- MRFP
            dx2 = torch.empty(Mc, hid, device=dev, dtype=F16)          # fp16 out: the epilogue's wide (row-major) path
            self._mm(dc1_16, WT(f"m{i}.fc2"), Mc, hid, C, out16=dx2, act=7, aux=s["x2"], ldaux=hid, rpg=1)
            self._wgrad(dc1_16, s["g16"], Mc, C, hid, grads, m.fc2.weight, m.fc2.bias)
            dx1_16 = torch.empty(Mc, hid, device=dev, dtype=F16)
            half = hid // 2
            dst = [self._direct(q) for q in (m.dw3.weight, m.dw3.bias, m.dw5.weight, m.dw5.bias)]
            dw3 = dst[0].view(half, 9) if dst[0] is not None else torch.empty(half, 9, device=dev, dtype=F32)
            db3 = dst[1] if dst[1] is not None else torch.empty(half, device=dev, dtype=F32)
            dw5 = dst[2].view(half, 25) if dst[2] is not None else torch.empty(half, 25, device=dev, dtype=F32)
            db5 = dst[3] if dst[3] is not None else torch.empty(half, device=dev, dtype=F32)
            npart = ctypes.c_long(0)
            lib.wc_mrfp_dwconv_parts(hs3, len(shapes), B, hid, ctypes.byref(npart))
            part = torch.empty(npart.value * hid * 26, device=dev, dtype=F32)
            lib.wc_mrfp_dwconv_bwd(L.ptr(dx2), 1, L.ptr(s["x1"]), L.ptr(self._b(m.dw3.weight).view(-1), F32),
                                   L.ptr(self._b(m.dw5.weight).view(-1), F32), None, L.ptr(dx1_16), L.ptr(dw3), L.ptr(db3), L.ptr(dw5),
                                   L.ptr(db5), L.ptr(part), INV, hs3, len(shapes), B, hid, L.stream())
            grads[id(m.dw3.weight)], grads[id(m.dw3.bias)] = dw3.view(m.dw3.weight.shape), db3
            grads[id(m.dw5.weight)], grads[id(m.dw5.bias)] = dw5.view(m.dw5.weight.shape), db5
            dc = torch.empty(Mc, C, device=dev, dtype=F32)
            self._mm(dx1_16, WT(f"m{i}.fc1"), Mc, C, hid, out32=dc, resid=dc1)
            self._wgrad(dx1_16, s["c16"], Mc, hid, C, grads, m.fc1.weight, m.fc1.bias)
            dc3 = dc
        return dc3, dvs, grads

    def _ow_grads(self, dow16, q16, M, n, ld, att, grads):
        """Weight / bias gradients of the stacked sampling_offsets | attention_weights GEMM: ONE split-K weight-gradient GEMM over
        the stacked columns, then one reduction job per Linear over its row range of the partials, written straight into that
        parameter's gradient (the bucket view when TrainStep opted in) -- no slicing / accumulate launches afterwards."""
        K = att.d_model
        tiles = ops.wgrad_tiles(n, K)
        ns = 1
        while ns * 2 * tiles <= _WGRAD_WGS and M // (ns * 2) >= 256:
            ns *= 2
        part, ns = ops.wgrad_partials(dow16, q16, M, n, K, lda=ld, slices=ns, bias=True)
        r0 = 0
        for lin in (att.sampling_offsets, att.attention_weights):
            rows = lin.weight.shape[0]
            gw, gb = self._direct(lin.weight), self._direct(lin.bias)
            dw = gw.view(rows, K) if gw is not None else torch.empty(rows, K, device=dow16.device, dtype=F32)
            db = gb if gb is not None else torch.empty(rows, device=dow16.device, dtype=F32)
            self._pending.append((part, dw, db, ns, rows, K, r0, n))
            grads[id(lin.weight)], grads[id(lin.bias)] = dw.view(lin.weight.shape), db
            r0 += rows

    def run_backward(self, ctx, dy):
        """backward() plus the tiny post-processing that needs the reduced partials: the stacked gradients are split into
        their two Linears, and gamma's gradient comes out of the output projection's un-gated weight gradient
        G = dv1^T o1:  dWop = diag(gamma) G,  dbop = gamma * s,  dgamma = rowsum(Wop * G) + bop * s  (s = dv1^T 1)."""
        self._gamma_jobs = []
        dc0, dvs, grads = self.backward(ctx, dy)
        for t, G, gsum in self._gamma_jobs:
            op = t.to_v.output_proj
            dst = [self._direct(q) for q in (op.weight, op.bias, t.gamma)]
            dW = dst[0] if dst[0] is not None else torch.empty_like(G)
            db = dst[1] if dst[1] is not None else torch.empty_like(gsum)
            dg = dst[2] if dst[2] is not None else torch.empty_like(gsum)
            L.lib().wc_cti_gate_grads(L.ptr(G, F32), L.ptr(gsum, F32), L.ptr(self._b(t.gamma), F32), L.ptr(self._b(op.weight), F32),
                                      L.ptr(self._b(op.bias), F32), L.ptr(dW, F32), L.ptr(db, F32), L.ptr(dg, F32), G.shape[0], G.shape[1],
                                      L.stream())
            grads[id(op.weight)], grads[id(op.bias)], grads[id(t.gamma)] = dW.view(op.weight.shape), db, dg
        self._gamma_jobs = None
        B, S, C = ctx["B"], ctx["S"], ctx["C"]
        return (dc0 * INV).view(B, S, C), [d * INV if d is not None else None for d in dvs], grads


class ComerFunction(torch.autograd.Function):
    """autograd bridge: (c0, 4 adapter maps, params...) -> fused rows (B*h*w, C); backward = ComerEngine.run_backward."""

    @staticmethod
    def forward(ctx, engine, shapes, hw, c0, m0, m1, m2, m3, *params):
        y, c = engine.forward(c0, shapes, hw, [m0, m1, m2, m3])
        ctx.engine, ctx.c, ctx.map_shapes = engine, c, [m.shape for m in (m0, m1, m2, m3)]
        return y

    @staticmethod
    def backward(ctx, dy):
        eng = ctx.engine
        dc0, dvs, grads = eng.run_backward(ctx.c, dy.contiguous())
        ctx.c = None
        out = []
        for p in eng.params():
            g = grads.get(id(p))
            if g is not None and p.grad is not None and g.data_ptr() == p.grad.data_ptr():
                g = None                                   # written straight into the caller's gradient buffer (see _direct)
            out.append(g.reshape(p.shape) if g is not None else None)
        return (None, None, None, dc0) + tuple(d.view(s) for d, s in zip(dvs, ctx.map_shapes)) + tuple(out)


class ComerTokensFunction(torch.autograd.Function):
    """The same with the four WeCLIP adapters inside the engine: (c0, 4 frozen f16 token tensors, params...) -> fused rows."""

    @staticmethod
    def forward(ctx, engine, shapes, hw, Lq, c0, x0, x1, x2, x3, *params):
        y, c = engine.forward(c0, shapes, hw, None, x16=[x0, x1, x2, x3], Lq=Lq)
        ctx.engine, ctx.c = engine, c
        return y

    @staticmethod
    def backward(ctx, dy):
        eng = ctx.engine
        dc0, _, grads = eng.run_backward(ctx.c, dy.contiguous())
        ctx.c = None
        out = []
        for p in eng.params():
            g = grads.get(id(p))
            if g is not None and p.grad is not None and g.data_ptr() == p.grad.data_ptr():
                g = None
            out.append(g.reshape(p.shape) if g is not None else None)
        return (None, None, None, None, dc0, None, None, None, None) + tuple(out)
