"""Pixel-adaptive refinement on MI355X -- same module API as reference WeCLIP_model/PAR.py.

`PAR(dilations, num_iter)`; `forward(imgs (b,3,H,W), masks (b,C,H',W')) -> (b,C,H',W')`; the
`kernel` (8,1,3,3) buffer is kept so reference checkpoints (which store `par.kernel`,
SURVEY.md §5) load with identical keys.  The arithmetic runs in csrc/par.hip.
"""
import os

import torch
import torch.nn as nn

from .. import _lib as L
from ..resize import bilinear_resize


def _one_hot_neighbour_kernel():
    # reference get_kernel(), PAR.py:10-24: tap k selects one of the 8 neighbours (TL..BR).
    k = torch.zeros(8, 1, 3, 3)
    for i, (r, c) in enumerate(((0, 0), (0, 1), (0, 2), (1, 0), (1, 2), (2, 0), (2, 1), (2, 2))):
        k[i, 0, r, c] = 1
    return k


class PAR(nn.Module):
    def __init__(self, dilations, num_iter):
        super().__init__()
        self.dilations = list(dilations)
        self.num_iter = int(num_iter)
        self.register_buffer("kernel", _one_hot_neighbour_kernel())
        self.w1, self.w2 = 0.3, 0.01   # fixed in csrc/par.hip as in PAR.py:36-37

    def affinity(self, imgs):
        """aff (b, 8*len(dilations), H, W): PAR.py:64-88."""
        L.require_gpu()
        imgs = imgs.float().contiguous()
        b, c, h, w = imgs.shape
        if c != 3:
            raise RuntimeError("PAR expects 3-channel images")
        aff = torch.empty(b, 8 * len(self.dilations), h, w, device=imgs.device, dtype=torch.float32)
        d = L.int_array(self.dilations)
        L.lib().wc_par_affinity(L.ptr(imgs, torch.float32, "imgs"), L.ptr(aff), b, h, w, d,
                                len(self.dilations), L.stream())
        return aff

    def forward(self, imgs, masks):
        """Dispatched as the registered custom op `torch.ops.weclip.par_forward` (torch_ops.py) for CUDA tensors."""
        if masks.is_cuda and not torch.cuda.is_current_stream_capturing():
            from .. import torch_ops  # noqa: F401  (registers the op)
            return torch.ops.weclip.par_forward(imgs, masks, self.dilations, self.num_iter)
        return self._forward_impl(imgs, masks)

    def _forward_impl(self, imgs, masks):
        L.require_gpu()
        masks = masks.float().contiguous()
        b, C, h, w = masks.shape
        if imgs.shape[0] != b:
            raise RuntimeError("PAR: batch of imgs and masks differ")
        imgs = imgs.float().contiguous()
        if imgs.shape[-2:] != masks.shape[-2:]:
            # PAR.py:67 -- F.interpolate(imgs, size=masks, bilinear, align_corners=True)
            imgs = bilinear_resize(imgs, (h, w), align_corners=True)
        T = 8 * len(self.dilations)
        # "fast" precision: the affinities live as 16-bit fixed-point pairs + a per-pixel scale between the sweeps (half
        # the bytes of the HBM-bound sweep; |rounding error| <= max weight * 7.7e-6, error-diffused: csrc/par.hip)
        from .. import config
        h16 = T == 48 and not config.exact() and config.par_q16
        # group so that aff + masks of a group stay (mostly) inside the 256 MiB Infinity Cache across the sweeps: measured
        # at 512x512, C = 3, 16 images (fast form, 35.7 MB per image): groups of 6 / 7 / 8 / 9 / 16 -> 2.27 / 2.28 / 2.24 /
        # 2.39 / 2.71 ms; the groups are balanced (16 images = 8 + 8, not 6 + 6 + 4)
        per_img = ((T // 2 + 1 if h16 else T) + 3 * C) * h * w * 4
        gmax = max(1, min(b, (288 << 20) // max(per_img, 1)))
        n_groups = -(-b // gmax)
        group = -(-b // n_groups)
        if os.environ.get("WECLIP_PAR_GROUP"):
            group = max(1, min(b, int(os.environ["WECLIP_PAR_GROUP"])))
        out = torch.empty_like(masks)
        tmp = torch.empty_like(masks)
        aff = torch.empty(group * T * h * ((w + 63) // 64 * 64), device=masks.device, dtype=torch.float32)
        d = L.int_array(self.dilations)
        fwd = L.lib().wc_par_forward_h if h16 else L.lib().wc_par_forward
        fwd(L.ptr(imgs, torch.float32, "imgs"), L.ptr(masks), L.ptr(out), L.ptr(tmp), L.ptr(aff), b, C, h, w, d,
            len(self.dilations), self.num_iter, group, L.stream())
        return out


def refine_labels(masks, valid_key, nch=None):
    """labels = valid_key[argmax_c masks] (reference `_refine_cams`, model_attn_aff_voc.py:49-57).
    masks (B,C,H,W) f32, valid_key (B,C) int64, nch optional (B,) int32."""
    L.require_gpu()
    B, C, H, W = masks.shape
    labels = torch.empty(B, H, W, device=masks.device, dtype=torch.int64)
    L.lib().wc_par_labels(L.ptr(masks, torch.float32, "masks"), L.ptr(valid_key, torch.int64, "valid_key"),
                          L.ptr(nch, torch.int32, "nch"), L.ptr(labels), B, C, H, W, L.stream())
    return labels
