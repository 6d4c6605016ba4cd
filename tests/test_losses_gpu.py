"""Fused segmentation loss (up-sampling + two CE terms) and the bilinear backward vs torch autograd."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import weclip_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,nc,h,w,scale", [(2, 21, 4, 6, 16), (1, 5, 7, 5, 16), (2, 21, 32, 32, 16),
                                           (1, 4, 5, 7, 9.5), (1, 3, 12, 10, 1), (1, 40, 4, 6, 16)])
def test_fused_seg_loss_forward_backward(B, nc, h, w, scale):
    from weclip_vit_comer_amd.utils.losses import get_seg_loss_fused
    g = torch.Generator().manual_seed(h)
    H, W = int(h * scale), int(w * scale)       # 9.5: non-integer ratio; 1: identity resize
    seg = torch.randn(B, nc, h, w, generator=g)
    lab = torch.randint(0, nc, (B, H, W), generator=g)
    lab[:, : H // 3] = 0
    lab[:, -5:, -7:] = 255
    ref_in = seg.double().requires_grad_(True)
    ref = O.seg_loss(F.interpolate(ref_in, size=(H, W), mode="bilinear", align_corners=False), lab)
    ref.backward()
    x = seg.cuda().requires_grad_(True)
    out = get_seg_loss_fused(x, lab.cuda())
    (3.0 * out).backward()
    assert abs(out.item() - ref.item()) < 2e-5 * max(1.0, abs(ref.item()))
    np.testing.assert_allclose(x.grad.cpu().numpy(), 3.0 * ref_in.grad.numpy(), rtol=2e-3, atol=2e-7)
    # without a gradient request the forward-only kernel runs (nc <= 24 with one: loss + gradient in one pixel pass)
    with torch.no_grad():
        out_ng = get_seg_loss_fused(seg.cuda(), lab.cuda())
    assert abs(out_ng.item() - out.item()) < 2e-6 * max(1.0, abs(out.item()))


@pytest.mark.parametrize("align", [False, True])
def test_bilinear_upsample_backward(align):
    from weclip_vit_comer_amd.resize import bilinear_upsample
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 3, 5, 7, generator=g)
    gy = torch.randn(2, 3, 40, 61, generator=g)
    xr = x.double().requires_grad_(True)
    (F.interpolate(xr, size=(40, 61), mode="bilinear", align_corners=align) * gy.double()).sum().backward()
    xc = x.cuda().requires_grad_(True)
    (bilinear_upsample(xc, (40, 61), align) * gy.cuda()).sum().backward()
    np.testing.assert_allclose(xc.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("B,h,w,radius", [(2, 4, 6, 8), (1, 32, 32, 8), (2, 10, 7, 2)])
def test_fused_affinity_loss_forward_backward(B, h, w, radius):
    """label -> affinity label (nearest /16, radius mask, ignore) + get_aff_loss, one pass, vs the oracle."""
    from weclip_vit_comer_amd.utils.losses import get_aff_loss_fused
    g = torch.Generator().manual_seed(h * w)
    H, W = 16 * h, 16 * w
    cam = torch.randint(0, 4, (B, H, W), generator=g)
    cam[:, :, : W // 5] = 255                      # ignored stripe
    ap = torch.rand(B, h * w, h * w, generator=g)
    ref_in = ap.double().requires_grad_(True)
    aff_label = O.cams_to_affinity_label(cam, O.radius_mask(h, w, radius), ignore_index=255)
    ref = O.aff_loss(ref_in, aff_label)
    ref = ref[0] if isinstance(ref, tuple) else ref
    ref.backward()
    x = ap.cuda().requires_grad_(True)
    out = get_aff_loss_fused(x, cam.cuda(), radius=radius, ignore_index=255)
    (0.1 * out).backward()
    assert abs(out.item() - ref.item()) < 1e-5
    np.testing.assert_allclose(x.grad.cpu().numpy(), 0.1 * ref_in.grad.numpy(), rtol=1e-4, atol=1e-10)


@pytest.mark.parametrize("B,nc,h,w,scale", [(2, 21, 4, 6, 16), (1, 81, 5, 3, 16), (2, 21, 32, 32, 16)])
def test_fused_seg_loss_backward_matches_the_two_kernel_path(B, nc, h, w, scale):
    """wc_seg_loss_bwd_fused (soft-max gradient formed inside the Y pass, no (B, nc, H, W) tensor) vs wc_seg_loss_bwd +
    wc_bilinear_resize_bwd: same per-pixel arithmetic, same summation order -> the same bits."""
    from weclip_vit_comer_amd import _lib as L
    H, W = h * scale, w * scale
    g = torch.Generator().manual_seed(B * nc + h)
    seg = torch.randn(B, nc, h, w, generator=g).cuda()
    lab = torch.randint(0, nc, (B, H, W), generator=g)
    lab[:, :3] = 255
    lab = lab.cuda()
    wts = torch.tensor([0.37e-4, 0.81e-4], device="cuda")
    F32 = torch.float32
    ghr = torch.empty(B, nc, H, W, device="cuda")
    L.lib().wc_seg_loss_bwd(L.ptr(seg, F32), L.ptr(lab, torch.int64), L.ptr(wts, F32), L.ptr(ghr), B, nc, h, w, H, W, 255, L.stream())
    ref, tmp = torch.empty_like(seg), torch.empty(B * nc * h * W, device="cuda")
    L.lib().wc_bilinear_resize_bwd(L.ptr(ghr), L.ptr(ref), L.ptr(tmp), B * nc, h, w, H, W, 0, L.stream())
    out, tmp2 = torch.empty_like(seg), torch.empty(B * nc * h * W, device="cuda")
    L.lib().wc_seg_loss_bwd_fused(L.ptr(seg, F32), L.ptr(lab, torch.int64), L.ptr(wts, F32), L.ptr(tmp2), L.ptr(out), B, nc, h, w, H, W,
                                  255, L.stream())
    assert ref.abs().max().item() > 0
    if nc > 24:
        assert torch.equal(out, ref)
    else:       # the small-class-count form takes the per-pixel maximum first instead of the online log-sum-exp: equal to rounding
        assert (out - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()
