"""ViT-CoMer feature-interaction inserts for WeCLIP.

There is no CoMer code in the reference repository -- only the paper (ViT_CoMer.pdf §3.2-3.3) and
the task brief (WeCLIP+ViT-CoMer.hwp): a CNN branch gives a {1/8, 1/16, 1/32} pyramid refined by MRFP
(FC -> multi-kernel depth-wise conv -> FC); at the four stage ends (ViT blocks [2,5,8,11]) a
bidirectional CTI exchanges features with the ViT branch through multi-scale deformable attention:
  CTI-toV:  v <- v + g * MSDeformAttn(LN(v), LN(c))          (query: ViT tokens, values: pyramid)
  CTI-toC:  c <- c + MSDeformAttn(LN(c), LN(v));  c <- c + FFN(LN(c))
Per the brief the ViT stays frozen and the *WeCLIP adapter outputs* (256-d maps of those blocks) are
the ViT-side features; the 8 CTI outputs (4 toV maps + 4 toC 1/16 maps) are channel-concatenated and
fused by a 1x1 conv into the decoder input.  Everything here is trainable.

Two implementations of the same arithmetic:
  * the ENGINE (comer_engine.py; the CUDA path in `fast` precision): everything behind the conv stem as one explicit
    forward / backward on fused HIP launches (csrc/comer.hip, msdeform.hip, gemm.hip), optionally including the four
    stage adapters (`forward_tokens`);
  * the MODULE-BY-MODULE form below (the CPU path, `exact` precision, `WECLIP_COMER_ENGINE=0`): the deformable-attention
    core on csrc/msdeform.hip, every Linear / 1x1 conv / LayerNorm through hip_functional.py (registered custom ops with
    autograd: MFMA GEMM forward, input / weight / bias gradients), depth-wise convs on csrc/dwconv.hip.
The 3x3 stride-2 stem convs + GroupNorm + ReLU (csrc/convstem.hip) are shared.  Parity is pinned only against
oracle/comer_oracle.py, stock torch modules and an fp64 evaluation of this network (no reference code exists).
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib as L
from ..hip_functional import module_layer_norm as _ln, module_linear as _lin

F32 = torch.float32


class _MSDAFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, value, loc, attn, shapes):
        value, loc, attn = value.float().contiguous(), loc.float().contiguous(), attn.float().contiguous()
        N, S, M, D = value.shape
        Lq, nL, P = loc.shape[1], loc.shape[3], loc.shape[4]
        out = torch.empty(N, Lq, M * D, device=value.device, dtype=F32)
        hs = L.int_array([v for hw in shapes for v in hw])
        L.lib().wc_msda_fwd(L.ptr(value, F32, "value"), hs, nL, L.ptr(loc, F32, "loc"), L.ptr(attn, F32, "attn"),
                            L.ptr(out), N, Lq, M, D, P, L.stream())
        ctx.save_for_backward(value, loc, attn)
        ctx.shapes = shapes
        return out

    @staticmethod
    def backward(ctx, gout):
        value, loc, attn = ctx.saved_tensors
        N, S, M, D = value.shape
        Lq, nL, P = loc.shape[1], loc.shape[3], loc.shape[4]
        gv, gl, ga = torch.empty_like(value), torch.empty_like(loc), torch.empty_like(attn)
        gmax = torch.empty(1, device=value.device, dtype=torch.int32)
        ws = torch.empty(N * M * (2 * S + nL * Lq * P * 8), device=value.device, dtype=torch.int32)   # bucket starts / sizes / (row, weight) entries
        hs = L.int_array([v for hw in ctx.shapes for v in hw])
        L.lib().wc_msda_bwd(L.ptr(value), hs, nL, L.ptr(loc), L.ptr(attn), L.ptr(gout.float().contiguous(), F32, "gout"),
                            L.ptr(gv), L.ptr(gl), L.ptr(ga), L.ptr(gmax), L.ptr(ws), N, Lq, M, D, P, L.stream())
        return gv, gl, ga, None


def ms_deform_attn_core(value, shapes, loc, attn):
    """value (N,S,M,D), shapes [(H,W)...], loc (N,Lq,M,nL,P,2), attn (N,Lq,M,nL,P) -> (N,Lq,M*D)."""
    L.require_gpu()
    return _MSDAFunction.apply(value, loc, attn, tuple(tuple(s) for s in shapes))


class MSDeformAttn(nn.Module):
    def __init__(self, d_model=256, n_levels=3, n_heads=8, n_points=4):
        super().__init__()
        self.d_model, self.n_levels, self.n_heads, self.n_points = d_model, n_levels, n_heads, n_points
        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)
        nn.init.zeros_(self.sampling_offsets.weight)
        th = torch.arange(n_heads, dtype=torch.float32) * (2.0 * math.pi / n_heads)
        grid = torch.stack([th.cos(), th.sin()], -1)
        grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(n_heads, 1, 1, 2).repeat(1, n_levels, n_points, 1)
        grid = grid * torch.arange(1, n_points + 1, dtype=torch.float32).view(1, 1, n_points, 1)
        with torch.no_grad():
            self.sampling_offsets.bias.copy_(grid.view(-1))
        nn.init.zeros_(self.attention_weights.weight)
        nn.init.zeros_(self.attention_weights.bias)

    def forward(self, query, reference_points, feat, shapes):
        """query (N,Lq,C); reference_points (N,Lq,nL,2) in [0,1]; feat (N,S,C), S = sum H_l*W_l."""
        N, Lq, C = query.shape
        M, nL, P = self.n_heads, self.n_levels, self.n_points
        value = _lin(self.value_proj, feat).view(N, feat.shape[1], M, C // M)
        off = _lin(self.sampling_offsets, query).view(N, Lq, M, nL, P, 2)
        aw = F.softmax(_lin(self.attention_weights, query).view(N, Lq, M, nL * P), -1).view(N, Lq, M, nL, P)
        norm = _level_sizes(tuple(tuple(hw) for hw in shapes), query.device)
        loc = reference_points[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
        return _lin(self.output_proj, ms_deform_attn_core(value, shapes, loc, aw))


_CONST = {}      # per (kind, shape, device) constants: built once, so a step neither copies from the host nor re-launches
                 # their arithmetic (and the step can be captured in a HIP graph)


def _level_sizes(shapes, device):
    key = ("sizes", shapes, str(device))
    if key not in _CONST:
        _CONST[key] = torch.tensor([[w, h] for h, w in shapes], dtype=torch.float32, device=device)
    return _CONST[key]


def _ref_points(h, w, device):
    key = ("ref", h, w, str(device))
    if key not in _CONST:
        ys, xs = torch.meshgrid((torch.arange(h, device=device) + 0.5) / h, (torch.arange(w, device=device) + 0.5) / w,
                                indexing="ij")
        _CONST[key] = torch.stack([xs.reshape(-1), ys.reshape(-1)], -1)        # (h*w, 2) as (x, y)
    return _CONST[key]


class _DWConvFunction(torch.autograd.Function):
    """nn.Conv2d(C, C, k, padding=k//2, groups=C) on the HIP kernels of csrc/dwconv.hip (MIOpen's grouped
    backward-weight path takes milliseconds on these small maps)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x = x.float().contiguous()
        w = weight.detach().float().contiguous()
        N, C, H, W = x.shape
        k = w.shape[-1]
        y = torch.empty_like(x)
        L.lib().wc_dwconv_fwd(L.ptr(x, torch.float32, "x"), L.ptr(w, torch.float32, "w"),
                              L.ptr(bias.detach().float().contiguous(), torch.float32, "bias") if bias is not None else None,
                              L.ptr(y), N, C, H, W, k, L.stream())
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.float().contiguous()
        N, C, H, W = x.shape
        k = w.shape[-1]
        dx, dw = torch.empty_like(x), torch.empty_like(w)
        db = torch.empty(C, device=x.device, dtype=torch.float32) if ctx.has_bias else None
        part = torch.empty(N * C * (k * k + 1), device=x.device, dtype=torch.float32)
        L.lib().wc_dwconv_bwd(L.ptr(x), L.ptr(w), L.ptr(dy, torch.float32, "dy"), L.ptr(dx), L.ptr(dw), L.ptr(db), L.ptr(part),
                              N, C, H, W, k, L.stream())
        return dx, dw, db


def _dwconv(conv, x):
    if x.is_cuda and conv.stride == (1, 1) and conv.dilation == (1, 1) and conv.groups == conv.in_channels == conv.out_channels \
            and conv.kernel_size[0] == conv.kernel_size[1] and conv.kernel_size[0] <= 7 and conv.padding == (conv.kernel_size[0] // 2,) * 2:
        return _DWConvFunction.apply(x, conv.weight, conv.bias)
    return conv(x)


class MRFP(nn.Module):
    """FC -> depth-wise convs with two receptive fields (3x3 / 5x5 on channel halves) -> FC."""

    def __init__(self, dim=256, hidden=128):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(dim, hidden), nn.Linear(hidden, dim)
        self.dw3 = nn.Conv2d(hidden // 2, hidden // 2, 3, padding=1, groups=hidden // 2)
        self.dw5 = nn.Conv2d(hidden // 2, hidden // 2, 5, padding=2, groups=hidden // 2)

    def forward(self, c, shapes):
        x = _lin(self.fc1, c)
        outs, s = [], 0
        for h, w in shapes:
            t = x[:, s:s + h * w].transpose(1, 2).reshape(x.shape[0], -1, h, w)
            a, b = t.chunk(2, dim=1)
            outs.append(torch.cat([_dwconv(self.dw3, a), _dwconv(self.dw5, b)], 1).flatten(2).transpose(1, 2))
            s += h * w
        return c + _lin(self.fc2, F.gelu(torch.cat(outs, 1)))


class CTI(nn.Module):
    def __init__(self, dim=256, heads=8, points=4):
        super().__init__()
        self.nv_q, self.nv_f = nn.LayerNorm(dim), nn.LayerNorm(dim)
        self.to_v = MSDeformAttn(dim, 3, heads, points)
        self.gamma = nn.Parameter(torch.zeros(dim))
        self.nc_q, self.nc_f = nn.LayerNorm(dim), nn.LayerNorm(dim)
        self.to_c = MSDeformAttn(dim, 1, heads, points)
        self.ffn_norm = nn.LayerNorm(dim)
        self.ffn = nn.Sequential(nn.Linear(dim, dim), nn.GELU(), nn.Linear(dim, dim))

    def forward(self, v, c, hw, shapes):
        """v (N, h*w, C) ViT-side tokens at 1/16; c (N, S, C) pyramid tokens."""
        h, w = hw
        dev = v.device
        rv = _ref_points(h, w, dev)[None, :, None, :].expand(v.shape[0], -1, 3, -1)
        v = v + self.gamma * self.to_v(_ln(self.nv_q, v), rv, _ln(self.nv_f, c), shapes)
        rc = torch.cat([_ref_points(a, b, dev) for a, b in shapes], 0)[None, :, None, :].expand(v.shape[0], -1, 1, -1)
        c = c + self.to_c(_ln(self.nc_q, c), rc, _ln(self.nc_f, v), [(h, w)])
        c = c + _lin(self.ffn[2], self.ffn[1](_lin(self.ffn[0], _ln(self.ffn_norm, c))))
        return v, c


class SpatialPrior(nn.Module):
    """Small conv stem giving the {1/8, 1/16, 1/32} pyramid at `dim` channels."""

    def __init__(self, dim=256, inplanes=32):
        super().__init__()
        def block(i, o, s):
            return nn.Sequential(nn.Conv2d(i, o, 3, s, 1, bias=False), nn.GroupNorm(8, o), nn.ReLU(inplace=True))
        self.stem = nn.Sequential(block(3, inplanes, 2), block(inplanes, inplanes, 2))          # 1/4
        self.c2 = block(inplanes, 2 * inplanes, 2)                                              # 1/8
        self.c3 = block(2 * inplanes, 4 * inplanes, 2)                                          # 1/16
        self.c4 = block(4 * inplanes, 4 * inplanes, 2)                                          # 1/32
        self.p2, self.p3, self.p4 = nn.Conv2d(2 * inplanes, dim, 1), nn.Conv2d(4 * inplanes, dim, 1), nn.Conv2d(4 * inplanes, dim, 1)

    def forward(self, img):
        if img.is_cuda:
            return self._forward_hip(img)
        x = self.stem(img)
        c2 = self.c2(x)
        c3 = self.c3(c2)
        c4 = self.c4(c3)
        shapes = [tuple(f.shape[-2:]) for f in (c2, c3, c4)]
        toks = [_lin(p, f.flatten(2).transpose(1, 2)) for p, f in ((self.p2, c2), (self.p3, c3), (self.p4, c4))]
        return torch.cat(toks, 1), shapes

    def _forward_hip(self, img):
        """The same five conv -> GroupNorm -> ReLU blocks on NHWC token rows: im2col + MFMA GEMM, fixed-order GroupNorm
        (hip_functional.conv3x3_rows / groupnorm_relu_rows); the 1x1 projections are GEMMs on the same rows."""
        from ..hip_functional import conv3x3_rows, groupnorm_relu_rows
        N, _, H, W = img.shape
        x = img.float().permute(0, 2, 3, 1).reshape(N * H * W, 3)
        feats = []
        for blk in (self.stem[0], self.stem[1], self.c2, self.c3, self.c4):
            conv, gn = blk[0], blk[1]
            x, H, W = conv3x3_rows(x, conv.weight, N, H, W, conv.stride[0])
            x = groupnorm_relu_rows(x, gn, N)
            feats.append((x, H, W))
        shapes = [(h, w) for _, h, w in feats[2:]]
        toks = [_lin(p, f.view(N, h * w, -1)) for p, (f, h, w) in zip((self.p2, self.p3, self.p4), feats[2:])]
        return torch.cat(toks, 1), shapes


class CoMerInteraction(nn.Module):
    """4 stages of MRFP + bidirectional CTI on the adapter outputs of ViT blocks `stage_blocks`;
    returns the 1x1-conv fusion of the 8 CTI outputs as the decoder input (B, dim, h, w)."""

    def __init__(self, dim=256, stage_blocks=(2, 5, 8, 10), heads=8, points=4):
        super().__init__()
        self.stage_blocks = tuple(stage_blocks)
        self.spm = SpatialPrior(dim)
        self.mrfp = nn.ModuleList([MRFP(dim) for _ in stage_blocks])
        self.cti = nn.ModuleList([CTI(dim, heads, points) for _ in stage_blocks])
        self.fuse = nn.Conv2d(2 * len(stage_blocks) * dim, dim, 1)
        self._engine = None
        # set by train_step.TrainStep: the engine may WRITE parameter gradients into `.grad` (views of the flat all-reduce
        # bucket, zeroed every step) instead of returning tensors that autograd adds to them one launch at a time
        self.direct_grads = False

    def _forward_engine(self, img, adapter_maps, hw):
        """CUDA path: the conv stem on its autograd Functions, everything behind it as ONE explicit forward / backward
        engine on fused HIP launches (comer_engine.py); same arithmetic as the module-by-module form below."""
        from ..comer_engine import ComerEngine, ComerFunction
        if self._engine is None:
            self._engine = ComerEngine(self)
        self._engine.adapters = None
        h, w = hw
        c0, shapes = self.spm(img)
        maps = [adapter_maps[b] for b in self.stage_blocks]
        y = ComerFunction.apply(self._engine, tuple(tuple(s) for s in shapes), (h, w), c0, *maps, *self._engine.params())
        return y.view(img.shape[0], h * w, -1).transpose(1, 2).reshape(img.shape[0], -1, h, w)

    def engine_ok(self, img):
        from .. import config
        return img.is_cuda and len(self.stage_blocks) == 4 and not config.exact() and os.environ.get("WECLIP_COMER_ENGINE", "1") != "0"

    def forward_tokens(self, img, x16, Lq, adapters, hw):
        """The engine path fed with the encoder's own f16 block outputs: x16[b] (B*Lq, Cin) f16 token rows (CLS first per
        image) of ViT block b, adapters[b] the WeCLIP adapter MLP of that block.  The four adapters of `stage_blocks` run inside
        the engine (two GEMMs each way instead of the module-by-module Linear ops)."""
        from ..comer_engine import ComerEngine, ComerTokensFunction
        if self._engine is None:
            self._engine = ComerEngine(self)
        eng = self._engine
        eng.adapters = [adapters[b] for b in self.stage_blocks]
        h, w = hw
        c0, shapes = self.spm(img)
        y = ComerTokensFunction.apply(eng, tuple(tuple(s) for s in shapes), (h, w), Lq, c0, *[x16[b].hi for b in self.stage_blocks],
                                      *eng.params())
        return y.view(img.shape[0], h * w, -1).transpose(1, 2).reshape(img.shape[0], -1, h, w)

    def forward(self, img, adapter_maps, hw):
        """adapter_maps: list of (B, h*w, dim) adapter outputs, one per ViT block; hw = (h, w)."""
        from .. import config
        if img.is_cuda and len(self.stage_blocks) == 4 and not config.exact() and os.environ.get("WECLIP_COMER_ENGINE", "1") != "0":
            return self._forward_engine(img, adapter_maps, hw)      # (`exact` precision: the module-by-module form, hi+lo operands)
        h, w = hw
        c, shapes = self.spm(img)
        outs = []
        for i, blk in enumerate(self.stage_blocks):
            c = self.mrfp[i](c, shapes)
            v, c = self.cti[i](adapter_maps[blk], c, hw, shapes)
            n16 = shapes[0][0] * shapes[0][1]
            outs += [v, c[:, n16:n16 + shapes[1][0] * shapes[1][1]]]
        cat = torch.cat(outs, 2)                                                        # (B, hw, 8*dim)
        y = _lin(self.fuse, cat)
        return y.transpose(1, 2).reshape(img.shape[0], -1, h, w)
