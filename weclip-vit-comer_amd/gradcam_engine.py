"""Block-12 forward + analytic GradCAM backward for a whole batch (csrc/gradcam.hip).

reference: CLIP.forward_last_layer (clip/model.py:407-429) differentiated by
pytorch_grad_cam.BaseCAM.forward (base_cam.py:62-108) once per class per image.  Here the
forward runs once per image and the backward once per (image, class) pair, all pairs batched.
"""
import torch

from . import _lib as L
from . import ops
from .ops import F16, F32, Split
from .clip import vit_engine as VE

def _logit_scale(clip_model):
    """exp(logit_scale) as a host float, read back from the device only when the (frozen) parameter changes
    (a per-step .item() is a device->host synchronisation that stalls the launch queue)."""
    p = clip_model.logit_scale
    key = (p.data_ptr(), p._version)
    cached = getattr(clip_model, "_logit_scale_host", None)
    if cached is None or cached[0] != key:
        cached = (key, float(p.detach().exp().item()))
        clip_model._logit_scale_host = cached
    return cached[1]


GRAD_SCALE = 4096.0   # power of two carried by gradients stored as fp16 hi/lo MFMA operands


class _BwdPack:
    """Transposed fp16 copies of the last block's weights (K-contiguous for dX = dY W)."""

    def __init__(self, blk):
        t = lambda w: ops.split_f16(w.detach().float().t().contiguous())
        self.pjT = t(blk.mlp.c_proj.weight)       # (4E, E): dz = dx2 @ Wproj
        self.fcT = t(blk.mlp.c_fc.weight)         # (E, 4E): da2 = du @ Wfc
        self.outT = t(blk.attn.out_proj.weight)   # (E, E):  do = g @ Wout
        self.w_in = blk.attn.in_proj_weight.detach().float().contiguous()   # (3E, E)
        self.ln2_w = blk.ln_2.weight.detach().float().contiguous()


class LastLayerState:
    """Everything the last block's forward leaves behind for B images."""

    def __init__(self, clip_model, x2, mean, keep, B, Lq):
        self.m, self.x2, self.mean, self.keep, self.B, self.L = clip_model, x2, mean, keep, B, Lq
        blk = clip_model.visual.transformer.resblocks[-1]
        if getattr(blk, "_bwd_pack", None) is None or blk._bwd_pack.w_in.device != x2.device:
            blk._bwd_pack = _BwdPack(blk)
        self.bw = blk._bwd_pack
        self.blk = blk

    # -- helpers ------------------------------------------------------------------------------
    def _head(self, text_hat, text_idx, n_text, pair_img, pair_cls, Tmax):
        v = self.m.visual
        B, Lq = self.B, self.L
        E, Ed = v.proj.shape
        P = pair_img.numel()
        dev = self.x2.device
        nchunk = (Lq + 15) // 16
        partial = torch.empty(B * nchunk * E, device=dev, dtype=F32)
        probs = torch.zeros(P, Tmax, device=dev, dtype=F32)
        df = torch.empty(P, E, device=dev, dtype=F32)
        L.lib().wc_cam_head(L.ptr(self.x2, F32), L.ptr(v.ln_post.weight.detach().float(), F32),
                            L.ptr(v.ln_post.bias.detach().float(), F32),
                            L.ptr(v.proj.detach().float().contiguous(), F32), L.ptr(text_hat, F32),
                            L.ptr(text_idx, torch.int32), L.ptr(n_text, torch.int32),
                            L.ptr(pair_img, torch.int32), L.ptr(pair_cls, torch.int32),
                            _logit_scale(self.m), L.ptr(partial), L.ptr(probs),
                            L.ptr(df), B, P, Lq, E, Ed, Tmax, L.stream())
        return probs, df

    def class_probs(self, text):
        """softmax class probabilities of every image against ONE set of text rows (T, Ed)."""
        dev = self.x2.device
        T = text.shape[0]
        that = (text / text.norm(dim=1, keepdim=True)).contiguous()
        idx = torch.arange(T, dtype=torch.int32, device=dev).repeat(self.B, 1).contiguous()
        nt = torch.full((self.B,), T, dtype=torch.int32, device=dev)
        img = torch.arange(self.B, dtype=torch.int32, device=dev)
        cls = torch.zeros(self.B, dtype=torch.int32, device=dev)
        probs, _ = self._head(that, idx, nt, img, cls, T)
        return probs

    def grad_cam(self, text_hat, text_idx, n_text, pair_img, pair_cls, Tmax):
        """All pairs at once.  text_hat: unit-norm text rows (R, Ed); text_idx (P,Tmax) int32 rows
        of text_hat per pair; n_text (P,), pair_img (P,), pair_cls (P,) int32 device tensors.
        Returns (cams (P, hw) f32 in [0,1], probs (P,Tmax), w (P,E) channel weights)."""
        lib = L.lib()
        k, bw = self.keep, self.bw
        B, Lq = self.B, self.L
        E = self.x2.shape[1]
        H = self.blk.attn.num_heads
        DH = E // H
        P = pair_img.numel()
        M = P * Lq
        dev = self.x2.device
        st = L.stream
        probs, df = self._head(text_hat, text_idx, n_text, pair_img, pair_cls, Tmax)
        v = self.m.visual
        # dx2 = gs * d/dx2 (through mean-pool + ln_post)
        dx2 = torch.empty(M, E, device=dev, dtype=F32)
        from . import config
        ex = config.exact()      # fast mode: gradients (scaled by 2^12) as single fp16 operands
        dx2s = Split(torch.empty(M, E, device=dev, dtype=F16), torch.empty(M, E, device=dev, dtype=F16) if ex else None)
        lib.wc_lnpost_bwd(L.ptr(df), L.ptr(self.x2), L.ptr(v.ln_post.weight.detach().float(), F32),
                          GRAD_SCALE, L.ptr(pair_img), L.ptr(dx2), L.ptr(dx2s.hi), L.ptr(dx2s.lo), P, Lq, E,
                          st())
        # MLP backward: du = (dx2 Wproj) * QuickGELU'(u);  da2 = du Wfc
        du = Split(torch.empty(M, 4 * E, device=dev, dtype=F16), torch.empty(M, 4 * E, device=dev, dtype=F16) if ex else None)
        ops.gemm(dx2s, bw.pjT, M, 4 * E, E, out16=du.hi, out16lo=du.lo, act=4, aux=k["u32"],
                 rowmap=pair_img, rpg=Lq, ldaux=4 * E)
        da2 = torch.empty(M, E, device=dev, dtype=F32)
        ops.gemm(du, bw.fcT, M, E, 4 * E, out32=da2)
        # dx1 = dx2 + LN2_bwd(da2), unscaled and rounded to fp16 like autograd does at myAtt.py:321
        g16 = torch.empty(M, E, device=dev, dtype=F16)
        lib.wc_ln2_bwd_add(L.ptr(da2), L.ptr(dx2), L.ptr(k["x1"]), L.ptr(bw.ln2_w), GRAD_SCALE,
                           L.ptr(pair_img), L.ptr(g16), P, Lq, E, st())
        do16 = torch.empty(M, E, device=dev, dtype=F16)
        ops.gemm(g16, bw.outT, M, E, E, out16=do16)           # fp16 GEMM, fp16 result (half linear bwd)
        # attention backward reduced to column sums of dq, dk, dv
        ws = torch.empty(4, P, H, Lq, device=dev, dtype=F32)
        c = torch.empty(P, 3 * E, device=dev, dtype=F32)
        lib.wc_attn_bwd_colsum(L.ptr(k["qkv"]), L.ptr(do16), L.ptr(k["o32"]), L.ptr(k["lse"]),
                               L.ptr(pair_img), L.ptr(ws[0]), L.ptr(ws[1]), L.ptr(ws[2]), L.ptr(ws[3]),
                               L.ptr(c), P, Lq, H, DH, st())
        w = torch.empty(P, E, device=dev, dtype=F32)
        ws2 = torch.empty(16 * P * E, device=dev, dtype=F32)
        lib.wc_rowvec_matmul(L.ptr(c), L.ptr(bw.w_in), L.ptr(w), L.ptr(ws2), P, 3 * E, E, 1.0 / (Lq - 1), st())
        cams = torch.empty(P, Lq - 1, device=dev, dtype=F32)
        lib.wc_cam_map(L.ptr(k["a32"]), L.ptr(w), L.ptr(pair_img), L.ptr(cams), P, Lq, E, st())
        return cams, probs, w


def last_layer_forward(clip_model, rows, B, Lq, want_mean=True):
    """rows (B*L, E) fp32 = output of block layers-1.  Runs the last block keeping what the
    analytic backward needs."""
    blk = clip_model.visual.transformer.resblocks[-1]
    keep = {}
    x2, mean = VE.run_block(blk.pack(), rows, B, Lq, want_mean=want_mean, keep=keep, tag=b"@vit_attn")
    return LastLayerState(clip_model, x2, mean, keep, B, Lq)
