"""Device-resident CAM refinement for a whole batch (csrc/affinity.hip): affinity weight,
Sinkhorn scale vectors, box masks, T_sym^2 refinement, up-sampling + background score.
Host side of reference clip/clip_tool.py:106-216 without the per-class numpy/cv2 round trips."""
import ctypes

import torch

from . import _lib as L
from .ops import F32

I32 = torch.int32


def _map_array(maps):
    arr = (ctypes.c_void_p * len(maps))(*[m.data_ptr() for m in maps])
    return arr


def seg_layer_keep(maps, n_last=6):
    """Layer selection of the seg-trans branch (clip_tool.py:158-167) for a batch: (B, n_last) f32 of 0/1.
    keep_l = [diff_l <= mean diff], diff_l = sum(seg - map_l[1:,1:]); the seg term is common to all layers, so the
    decision is taken on A_l = sum(map_l[1:,1:]) alone, reduced in a fixed order (csrc/affinity.hip)."""
    sel = maps[-n_last:]
    B, Lq, _ = sel[0].shape
    hw = Lq - 1
    dev = sel[0].device
    for m in sel:
        L.ptr(m, F32, "attention map")
    diff = torch.empty(B, len(sel), device=dev, dtype=F32)
    wgt = torch.empty(B, len(sel), device=dev, dtype=F32)
    rowsum = torch.empty(B, len(sel), hw, device=dev, dtype=F32)
    L.lib().wc_aff_seg_weights(_map_array(sel), len(sel), L.ptr(rowsum), L.ptr(diff), L.ptr(wgt), B, Lq, L.stream())
    return (wgt > 0).float()


def fused_ok(hw):
    """The fused sweeps of csrc/affinity.hip need 16-byte row pieces (hw % 4 == 0) and hw <= 2048."""
    return hw % 4 == 0 and 4 <= hw <= 2048


def _nblk(hw):
    return (hw + 31) // 32


def affinity_weight(maps, seg=None, seg_trans=False, n_last=6, keep=None, return_c1=False):
    """maps: list of head-mean attention maps (B,L,L) f32 in layer order (11 encoder + last block).
    Normal branch (clip_tool.py:169-173): mean of the last 8 [1:,1:].  Seg-trans branch
    (:152-168, n_last 6 VOC / 10 COCO): masked mean of the last n_last times seg (B,hw,hw).
    keep (B, n_last) 0/1, optional: a caller-supplied layer selection instead of seg_layer_keep's."""
    sel = maps[-n_last:] if seg_trans else maps[-8:]      # entries outside the selection may be None
    B, Lq, _ = sel[0].shape
    hw = Lq - 1
    dev = sel[0].device
    for m in sel:
        L.ptr(m, F32, "attention map")
    lib = L.lib()
    W = torch.empty(B, hw, hw, device=dev, dtype=F32)

    def weight(wgt, segp):
        """W (and, on the fused path, the first Sinkhorn column scale c1 = 1 / colsum(W) from the same pass)."""
        if return_c1 and fused_ok(hw):
            c1 = torch.empty(B, hw, device=dev, dtype=F32)
            ws = torch.empty(B * ((hw + 7) // 8) * hw, device=dev, dtype=F32)
            lib.wc_aff_weight_c1(_map_array(sel), len(sel), L.ptr(wgt), L.ptr(segp), L.ptr(W), L.ptr(c1), L.ptr(ws), B, Lq,
                                 L.stream())
            return W, c1
        lib.wc_aff_weight(_map_array(sel), len(sel), L.ptr(wgt), L.ptr(segp), L.ptr(W), B, Lq, L.stream())
        return (W, None) if return_c1 else W

    if not seg_trans:
        wgt = torch.full((B, len(sel)), 1.0 / len(sel), device=dev, dtype=F32)
        return weight(wgt, None)
    seg = seg.detach().float().contiguous()
    if keep is not None:
        keep = keep.to(dev).float()
        wgt = (keep / (keep.sum(1, keepdim=True) + 1e-5)).contiguous()
    else:
        diff = torch.empty(B, len(sel), device=dev, dtype=F32)
        wgt = torch.empty(B, len(sel), device=dev, dtype=F32)
        rowsum = torch.empty(B, len(sel), hw, device=dev, dtype=F32)
        lib.wc_aff_seg_weights(_map_array(sel), len(sel), L.ptr(rowsum), L.ptr(diff), L.ptr(wgt), B, Lq, L.stream())
    return weight(wgt, seg)


def _matvec(W, X, sin=None, sout=None, add=None, transpose=False, recip=False, alpha=1.0):
    B, hw, _ = W.shape
    K = X.shape[-1]
    out = torch.empty(B, hw, K, device=W.device, dtype=F32)
    L.lib().wc_matvec(L.ptr(W, F32), L.ptr(X, F32), L.ptr(sin, F32), L.ptr(sout, F32), L.ptr(add, F32),
                      L.ptr(out), B, hw, K, 1 if transpose else 0, 1 if recip else 0, alpha, L.stream())
    return out


def sinkhorn_scales(W, rounds=3, c1=None):
    """Row/column scale vectors of the 3x (col-normalise, row-normalise) of compute_trans_mat
    (clip_tool.py:67-72): T = diag(r) W diag(c).  hw % 4 == 0: fused sweeps (a row pass and the next column pass
    share one read of W; `c1`, if given, is the first column scale that affinity_weight already produced)."""
    B, hw, _ = W.shape
    if fused_ok(hw):
        lib = L.lib()
        ws = torch.empty(B * _nblk(hw) * hw, device=W.device, dtype=F32)
        if c1 is None:
            c1 = _matvec(W, torch.ones(B, hw, 1, device=W.device, dtype=F32), transpose=True, recip=True).view(B, hw)
        c = c1
        for it in range(rounds):
            r = torch.empty(B, hw, device=W.device, dtype=F32)
            last = it == rounds - 1
            cn = None if last else torch.empty(B, hw, device=W.device, dtype=F32)
            lib.wc_aff_sinkhorn_step(L.ptr(W, F32), L.ptr(c, F32), L.ptr(r), L.ptr(cn), L.ptr(ws), B, hw, 1 if last else 0,
                                     L.stream())
            if not last:
                c = cn
        return r, c
    r = torch.ones(B, hw, 1, device=W.device, dtype=F32)
    c = None
    for _ in range(rounds):
        c = _matvec(W, r, transpose=True, recip=True)     # c = 1 / (W^T r)
        r = _matvec(W, c, transpose=False, recip=True)    # r = 1 / (W c)
    return r.view(B, hw), c.view(B, hw)


def tsym_apply(W, r, c, X):
    """T_sym X with T_sym = (diag(r) W diag(c) + diag(c) W^T diag(r)) / 2  (clip_tool.py:73)."""
    B, hw, _ = W.shape
    K = X.shape[-1]
    if fused_ok(hw):
        out = torch.empty(B, hw, K, device=W.device, dtype=F32)
        for k0 in range(0, K, 4):              # 4 classes per sweep
            kn = min(4, K - k0)
            Xs = X if (k0 == 0 and kn == K) else X[:, :, k0:k0 + kn].contiguous()
            os_ = out if (k0 == 0 and kn == K) else torch.empty(B, hw, kn, device=W.device, dtype=F32)
            y1 = torch.empty(B, hw, kn, device=W.device, dtype=F32)
            ws = torch.empty(B * _nblk(hw) * hw * kn, device=W.device, dtype=F32)
            L.lib().wc_aff_tsym_apply(L.ptr(W, F32), L.ptr(r.contiguous(), F32), L.ptr(c.contiguous(), F32), L.ptr(Xs, F32),
                                      L.ptr(os_), L.ptr(y1), L.ptr(ws), B, hw, kn, L.stream())
            if os_ is not out:
                out[:, :, k0:k0 + kn] = os_
        return out
    half = _matvec(W, X, sin=c, sout=r, alpha=0.5)
    return _matvec(W, X, sin=r, sout=c, add=half, transpose=True, alpha=0.5)


def trans_mat(W):
    """Public compute_trans_mat: materialised T_sym @ T_sym (clip_tool.py:64-80)."""
    B, hw, _ = W.shape
    r, c = sinkhorn_scales(W)
    T = torch.empty(B, hw, hw, device=W.device, dtype=F32)
    L.lib().wc_tsym(L.ptr(W, F32), L.ptr(r), L.ptr(c), L.ptr(T), B, hw, L.stream())
    out = torch.empty_like(T)
    for k0 in range(0, hw, 64):      # T @ T column block by column block through the mat-vec kernels
        X = T[:, :, k0:k0 + 64].contiguous()
        out[:, :, k0:k0 + 64] = _matvec(T, X)
    return out


def box_masks(cams, pair_img, pair_slot, B, K, h, w, thr, want_mask=False, want_boxes=False):
    """cams (P, h*w) f32 -> V (B, hw, K) with V[img, :, slot] = boxmask * cam."""
    P = cams.shape[0]
    dev = cams.device
    V = torch.zeros(B, h * w, K, device=dev, dtype=F32)
    mask = torch.empty(P, h * w, device=dev, dtype=F32) if want_mask else None
    boxes = torch.zeros(P, 64, 4, device=dev, dtype=I32) if want_boxes else None
    nbox = torch.zeros(P, device=dev, dtype=I32) if want_boxes else None
    L.lib().wc_box_mask(L.ptr(cams, F32, "cams"), L.ptr(pair_img, I32), L.ptr(pair_slot, I32), L.ptr(V),
                        L.ptr(mask), L.ptr(boxes), L.ptr(nbox), 64, P, h, w, K, float(thr), L.stream())
    return V, mask, boxes, nbox


def refine(W, cams, pair_img, pair_slot, K, h, w, thr, c1=None):
    """R (B, hw, K) = T_sym^2 (boxmask * cam) for every pair (clip_tool.py:179-191)."""
    B = W.shape[0]
    r, c = sinkhorn_scales(W, c1=c1)
    V, _, _, _ = box_masks(cams, pair_img, pair_slot, B, K, h, w, thr)
    return tsym_apply(W, r, c, tsym_apply(W, r, c, V))


def upsample_with_bg(R, nk, h, w, H, Wd, C=None):
    """generate_cam_label + bg score (clip_tool.py:202-216, model_attn_aff_voc.py:160-163):
    cams (B, C, H, W): channel 0 = 1 - max_k, channels 1.. = bilinear(min-max(R_k))."""
    B, hw, K = R.shape
    C = K + 1 if C is None else C
    stats = torch.empty(B, K, 2, device=R.device, dtype=F32)
    cams = torch.empty(B, C, H, Wd, device=R.device, dtype=F32)
    L.lib().wc_cam_upsample(L.ptr(R, F32), L.ptr(nk, I32), L.ptr(stats), L.ptr(cams), B, h, w, K, C, H, Wd,
                            L.stream())
    return cams
