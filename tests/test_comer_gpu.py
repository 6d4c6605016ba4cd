"""ViT-CoMer inserts (row a-9): HIP multi-scale deformable attention vs the CPU restatement
(oracle/comer_oracle.py; no reference code exists -> parity unpinned w.r.t. the reference)."""
import numpy as np
import pytest
import torch

from oracle import comer_oracle as CO

pytestmark = pytest.mark.gpu


def _inputs(N, shapes, Lq, M, D, P, seed=0):
    g = torch.Generator().manual_seed(seed)
    S = sum(h * w for h, w in shapes)
    value = torch.randn(N, S, M, D, generator=g)
    loc = torch.rand(N, Lq, M, len(shapes), P, 2, generator=g) * 1.2 - 0.1      # some samples fall outside
    attn = torch.softmax(torch.randn(N, Lq, M, len(shapes) * P, generator=g), -1).view(N, Lq, M, len(shapes), P)
    return value, loc, attn


@pytest.mark.parametrize("shapes,Lq,M,D,P", [([(8, 12), (4, 6), (2, 3)], 24, 8, 32, 4), ([(5, 7)], 50, 4, 16, 2),
                                             ([(64, 64), (32, 32), (16, 16)], 1024, 8, 32, 4)])
def test_msda_forward_backward_vs_oracle(shapes, Lq, M, D, P):
    from weclip_vit_comer_amd.WeCLIP_model.comer import ms_deform_attn_core
    N = 2
    value, loc, attn = _inputs(N, shapes, Lq, M, D, P)
    vr, lr, ar = [t.double().requires_grad_(True) for t in (value, loc, attn)]
    ref = CO.ms_deform_attn(vr, shapes, lr, ar)
    g = torch.randn(ref.shape, generator=torch.Generator().manual_seed(1), dtype=torch.float64)
    (ref * g).sum().backward()
    vg, lg, ag = [t.cuda().requires_grad_(True) for t in (value, loc, attn)]
    out = ms_deform_attn_core(vg, shapes, lg, ag)
    (out * g.float().cuda()).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(vg.grad.cpu().numpy(), vr.grad.numpy(), rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(ag.grad.cpu().numpy(), ar.grad.numpy(), rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(lg.grad.cpu().numpy(), lr.grad.numpy(), rtol=1e-3, atol=2e-3)


def test_comer_interaction_module_trains():
    from weclip_vit_comer_amd.WeCLIP_model.comer import CoMerInteraction
    torch.manual_seed(0)
    B, H, W, dim = 2, 64, 96, 256
    net = CoMerInteraction(dim).cuda()
    img = torch.randn(B, 3, H, W, device="cuda")
    maps = [torch.randn(B, (H // 16) * (W // 16), dim, device="cuda") for _ in range(11)]
    y = net(img, maps, (H // 16, W // 16))
    assert tuple(y.shape) == (B, dim, H // 16, W // 16) and torch.isfinite(y).all()
    y.square().mean().backward()
    gs = [p.grad for p in net.parameters() if p.requires_grad]
    assert all(g is not None and torch.isfinite(g).all() for g in gs)
    assert net.cti[0].to_c.value_proj.weight.grad.abs().sum() > 0


def test_weclip_with_comer_inserts_runs_a_train_step():
    from oracle import synth
    from weclip_vit_comer_amd.WeCLIP_model.model_attn_aff_voc import WeCLIP
    from weclip_vit_comer_amd.train_step import TrainStep
    torch.manual_seed(0)
    sd = synth.make_clip_state_dict(**synth.TINY)
    bg, fg = synth.make_text_features(20, 25, synth.TINY["embed_dim"])
    m = WeCLIP(num_classes=21, clip_model=sd, embedding_dim=256, in_channels=[64] * 4, device="cuda",
               text_features=(bg.cuda(), fg.cuda()), comer=True)
    assert any(k.startswith("comer.") for k in m.state_dict())
    step = TrainStep(m)
    img = synth.make_images(2, *synth.TINY_HW).cuda()
    before = m.comer.fuse.weight.detach().clone()
    loss, ls, la = step(img, labels=synth.TINY_LABELS)
    assert torch.isfinite(loss) and not torch.equal(before, m.comer.fuse.weight.detach())


@pytest.mark.parametrize("N,C,H,W,k", [(2, 64, 17, 23, 3), (1, 8, 64, 64, 5), (3, 5, 6, 4, 7)])
def test_depthwise_conv_matches_torch(N, C, H, W, k):
    """csrc/dwconv.hip (MRFP's nn.Conv2d(groups=C)) forward and all three gradients vs torch's conv2d in fp64 on the CPU."""
    import torch.nn as nn
    from weclip_vit_comer_amd.WeCLIP_model.comer import _dwconv
    torch.manual_seed(k)
    conv = nn.Conv2d(C, C, k, padding=k // 2, groups=C)
    x = torch.randn(N, C, H, W)
    ref_conv = nn.Conv2d(C, C, k, padding=k // 2, groups=C).double()
    ref_conv.load_state_dict({n: v.double() for n, v in conv.state_dict().items()})
    xr = x.double().requires_grad_(True)
    yr = ref_conv(xr)
    gy = torch.randn(N, C, H, W)
    yr.backward(gy.double())
    conv = conv.cuda()
    xg = x.cuda().requires_grad_(True)
    y = _dwconv(conv, xg)
    y.backward(gy.cuda())
    assert (y.detach().cpu().double() - yr.detach()).abs().max().item() < 1e-5
    assert (xg.grad.cpu().double() - xr.grad).abs().max().item() < 1e-5
    assert (conv.weight.grad.cpu().double() - ref_conv.weight.grad).abs().max().item() < 1e-4 * ref_conv.weight.grad.abs().max().item()
    assert (conv.bias.grad.cpu().double() - ref_conv.bias.grad).abs().max().item() < 1e-4 * ref_conv.bias.grad.abs().max().item()


def test_comer_hip_decoder_matches_stock_autograd_decoder():
    """With the inserts enabled the decoder / linear_pred / attn_pred run on the HIP path from the fused features
    (DecoderFunction); outputs and the gradients reaching the inserts must match the stock-autograd decoder."""
    from oracle import synth
    from weclip_vit_comer_amd.WeCLIP_model.model_attn_aff_voc import WeCLIP
    from weclip_vit_comer_amd.utils.losses import get_seg_loss_fused
    img = synth.make_images(2, *synth.TINY_HW).cuda()
    res = {}
    for impl in ("hip", "torch"):
        torch.manual_seed(0)
        sd = synth.make_clip_state_dict(**synth.TINY)
        bg, fg = synth.make_text_features(20, 25, synth.TINY["embed_dim"])
        m = WeCLIP(num_classes=21, clip_model=sd, embedding_dim=256, in_channels=[64] * 4, device="cuda",
                   text_features=(bg.cuda(), fg.cuda()), comer=True)
        m.eval()
        m.head_impl = impl
        seg, lab, ap = m(img, ["a", "b"], labels=synth.TINY_LABELS)
        loss = get_seg_loss_fused(seg, lab) + 0.1 * ap.mean()
        loss.backward()
        res[impl] = (seg.detach(), ap.detach(), m.comer.fuse.weight.grad.clone(),
                     m.decoder.linear_pred.weight.grad.clone(), m.comer.cti[0].to_v.value_proj.weight.grad.clone())
    for a, b, tol in zip(res["hip"], res["torch"], (5e-3, 5e-3, 5e-2, 5e-2, 5e-2)):
        assert (a - b).abs().max().item() <= tol * max(b.abs().max().item(), 1e-6), (a - b).abs().max().item()
