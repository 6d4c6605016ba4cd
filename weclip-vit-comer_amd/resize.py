"""Bilinear resize of fp32 NCHW tensors on the HIP path (csrc/resize.hip)."""
import torch

from . import _lib as L


def bilinear_resize(x, size, align_corners=False):
    """== F.interpolate(x, size, mode='bilinear', align_corners=...) for 4-D fp32 input."""
    L.require_gpu()
    x = x.float().contiguous()
    n, c, hs, ws = x.shape
    hd, wd = int(size[0]), int(size[1])
    out = torch.empty(n, c, hd, wd, device=x.device, dtype=torch.float32)
    L.lib().wc_bilinear_resize(L.ptr(x, torch.float32, "x"), L.ptr(out), n * c, hs, ws, hd, wd,
                               1 if align_corners else 0, L.stream())
    return out
