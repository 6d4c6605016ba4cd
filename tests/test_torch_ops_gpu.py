"""`torch.ops.weclip.*` (torch_ops.py) called for real on the MI355X, against the oracle / stock torch."""
import pytest
import torch

from oracle import synth
from oracle import weclip_oracle as O

pytestmark = pytest.mark.gpu


def test_registered_ops_run_the_hip_kernels():
    import weclip_vit_comer_amd as W
    W.register_torch_ops()
    g = torch.Generator().manual_seed(3)
    img = synth.make_images(1, 64, 96, seed=1)
    masks = torch.rand(1, 3, 64, 96, generator=g)
    out = torch.ops.weclip.par_forward(img.cuda(), masks.cuda(), [1, 2, 4, 8, 12, 24], 20)
    assert (out.cpu() - O.par(img, masks)).abs().max().item() < 5e-4
    vk = torch.tensor([[0, 4, 8]], dtype=torch.int64)
    lab = torch.ops.weclip.par_labels(out, vk.cuda())
    assert torch.equal(lab.cpu(), vk[0][out.cpu().argmax(1)])
    w = torch.rand(24, 24, generator=g) + 0.05
    T = torch.ops.weclip.trans_mat(w.cuda())
    assert (T.cpu() - O.compute_trans_mat(w)).abs().max().item() < 1e-5
    x = torch.randn(300, 128, generator=g)
    wt = torch.randn(40, 128, generator=g) * 0.1
    b = torch.randn(40, generator=g)
    y = torch.ops.weclip.linear_f16(x.half().cuda(), wt.half().cuda(), b.cuda(), 2)
    ref = torch.relu(x.half().double() @ wt.half().double().t() + b.double())
    assert (y.cpu().double() - ref).abs().max().item() < 1e-4
    ln = torch.ops.weclip.layernorm(x.cuda(), torch.ones(128).cuda(), torch.zeros(128).cuda(), 1e-5)
    assert (ln.cpu() - torch.nn.functional.layer_norm(x, (128,))).abs().max().item() < 1e-5
    r = torch.ops.weclip.bilinear_resize(masks.cuda(), 32, 40, False)
    assert (r.cpu() - torch.nn.functional.interpolate(masks, (32, 40), mode="bilinear", align_corners=False)).abs().max().item() < 1e-5
    lt = torch.randint(0, 21, (64, 96), generator=g)
    lp = torch.randint(0, 21, (64, 96), generator=g)
    h = torch.ops.weclip.confusion_hist(lt.cuda(), lp.cuda(), 21)
    assert int(h.sum()) == 64 * 96 and int(h[3, 5]) == int(((lt == 3) & (lp == 5)).sum())
    # the nn.Module goes through the registered op: it shows up under its own name in the profiler
    from weclip_vit_comer_amd.WeCLIP_model.PAR import PAR
    par = PAR([1, 2, 4, 8, 12, 24], 20).cuda()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU]) as prof:
        par(img.cuda(), masks.cuda())
    assert any("weclip::par_forward" in e.key for e in prof.key_averages())


def test_device_input_pipeline_matches_reference_fixture(golden):
    """f-1: csrc/augment.hip against the fixture made by the reference's own transforms (datasets/transforms.py: PIL BILINEAR
    random_scaling, random_fliplr, random_crop, normalize_img; real Pillow), with the recorded draws fed through the host
    side of `DeviceAugment` -- EQUAL on every pixel, up- and down-scaling (scales 0.52 ... 1.53)."""
    import numpy as np
    from weclip_vit_comer_amd.data import DeviceAugment
    g = golden("augment_ref.npz")
    f = synth.make_images(6, 54, 76, seed=700)
    imgs = (f * 58.0 + 118.0).clamp_(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
    assert synth.checksum([imgs]) == g["img_ck"]
    crop = int(g["crop"])
    aug = DeviceAugment(crop_size=crop, rescale_range=(0.5, 2.0), seed=int(g["seed"]))
    params = aug.draw(6, 54, 76)
    d = g["draws"]
    assert params[:, 4:].tolist() == d[:, 2:].astype(int).tolist() and params[:, 1].tolist() == (d[:, 1] > 0.5).astype(int).tolist()
    out = aug(imgs.cuda(), params).cpu().numpy()
    diff = np.abs(out - g["out"])
    print(f"device input pipeline vs the reference's transforms: max abs diff {diff.max():.3e} over {out.size} values")
    assert np.array_equal(out, g["out"]), (diff.max(), (diff > 0).mean())


def test_device_input_pipeline_matches_oracle_at_bench_size():
    """The loader's real geometry (375 x 500 uint8 sources, crop 512, scales 0.5 ... 2.0) against the oracle's restatement
    (pinned to Pillow and to the reference fixture on the CPU side): equal on every pixel."""
    import numpy as np
    from weclip_vit_comer_amd.data import DeviceAugment
    g = torch.Generator().manual_seed(9)
    imgs = torch.randint(0, 256, (4, 375, 500, 3), generator=g, dtype=torch.uint8)
    aug = DeviceAugment(crop_size=512, rescale_range=(0.5, 2.0), seed=5)
    draws = [aug.draw_one(375, 500) for _ in range(4)]
    draws[0] = (0.5,) + draws[0][1:2] + (187, 250) + (10, 20, 0, 0)          # the range's lower end: 5 x 5 taps
    out = aug(imgs.cuda(), aug.pack(draws)).cpu()
    assert tuple(out.shape) == (4, 3, 512, 512)
    for b, (s, flip, rh, rw, pad_y, pad_x, crop_y, crop_x) in enumerate(draws):
        ref = O.augment_normalize(imgs[b], s, flip, pad_y, pad_x, crop_y, crop_x, 512)
        assert (rh, rw) == (int(s * 375), int(s * 500))
        assert torch.equal(out[b], ref), (b, s, (out[b] - ref).abs().max().item())
    assert aug(imgs.cuda()).shape == out.shape      # fresh draw path
    with pytest.raises(RuntimeError):
        aug(imgs.cuda(), aug.pack([(0.2, 0, 75, 100, 0, 0, 0, 0)] * 4))      # beyond the 4x down-scaling the tables hold
    # device-resident params skip the host check: the C ABI's own precondition check poisons the pixels (never a truncated filter)
    bad = aug(imgs.cuda(), aug.pack([(0.2, 0, 75, 100, 0, 0, 0, 0)] * 4).cuda())
    # (at the image border Pillow clips the filter's support, which may leave <= 9 taps: those pixels stay exact and finite)
    assert torch.isnan(bad[:, :, 4:71, 4:96]).all() and not torch.isnan(bad[:, :, 75:, :]).any() and not torch.isnan(bad[:, :, :, 100:]).any()


@pytest.mark.parametrize("case", ["dx_only", "dw_only", "both"])
def test_trainable_linear_partial_gradients(case):
    """weclip::linear through register_autograd with a frozen weight (dx only), a frozen input (dW / db only) and both:
    the backward op hands back one placeholder PER unused slot (outputs of a custom op may not alias each other)."""
    import weclip_vit_comer_amd as W
    W.register_torch_ops()
    g = torch.Generator().manual_seed(11)
    x = torch.randn(512, 128, generator=g)
    wt = torch.randn(40, 128, generator=g) * 0.1
    b = torch.randn(40, generator=g)
    dy = torch.randn(512, 40, generator=g)
    xc = x.cuda().requires_grad_(case != "dw_only")
    wc = wt.cuda().requires_grad_(case != "dx_only")
    bc = b.cuda().requires_grad_(case != "dx_only")
    y = torch.ops.weclip.linear(xc, wc, bc, 2)
    y.backward(dy.cuda())
    # fp64 reference through the ReLU mask of the HIP forward (a pre-activation within fp16 rounding of 0 may land on either side)
    xr = x.double().requires_grad_(case != "dw_only")
    wr = wt.double().requires_grad_(case != "dx_only")
    br = b.double().requires_grad_(case != "dx_only")
    ((xr @ wr.t() + br) * (y.detach().cpu() > 0)).backward(dy.double())
    for name, got, ref in (("dx", xc.grad, xr.grad), ("dw", wc.grad, wr.grad), ("db", bc.grad, br.grad)):
        if ref is None:
            assert got is None, name
            continue
        err = (got.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
        assert err < 3e-3, (case, name, err)
