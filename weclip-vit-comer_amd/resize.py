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


class _BilinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, size, align_corners):
        ctx.meta = (tuple(x.shape), size, align_corners)
        return bilinear_resize(x, size, align_corners)

    @staticmethod
    def backward(ctx, g):
        (n, c, hs, ws), (hd, wd), ac = ctx.meta
        g = g.float().contiguous()
        out = torch.empty(n, c, hs, ws, device=g.device, dtype=torch.float32)
        tmp = torch.empty(n * c * hs * wd, device=g.device, dtype=torch.float32)
        L.lib().wc_bilinear_resize_bwd(L.ptr(g, torch.float32, "grad"), L.ptr(out), L.ptr(tmp), n * c, hs, ws, hd,
                                       wd, 1 if ac else 0, L.stream())
        return out, None, None


def bilinear_upsample(x, size, align_corners=False):
    """Differentiable bilinear up-sampling (forward csrc/resize.hip, backward a gather kernel)."""
    return _BilinearFn.apply(x, (int(size[0]), int(size[1])), align_corners)
