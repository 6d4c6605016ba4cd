"""Affinity / refinement kernels (K9-K11) on the HIP path vs the CPU oracle and reference goldens."""
import numpy as np
import pytest
import torch

from oracle import synth
from oracle import weclip_oracle as O

pytestmark = pytest.mark.gpu
I32 = torch.int32


@pytest.fixture(scope="module")
def P():
    from weclip_vit_comer_amd import cam_pipeline
    return cam_pipeline


def _maps(B, L, n=12, seed=0):
    g = torch.Generator().manual_seed(seed)
    ms = []
    for _ in range(n):
        s = torch.randn(B, L, L, generator=g) * 2
        ms.append(torch.softmax(s, -1))
    return ms


def test_trans_mat_matches_reference_golden(P, golden):
    g = golden("tiny_func.npz")
    W = torch.from_numpy(g["trans_in"])[None].cuda()
    out = P.trans_mat(W)[0].cpu().numpy()
    np.testing.assert_allclose(out, g["trans_out"], rtol=2e-5, atol=1e-9)


@pytest.mark.parametrize("seg_trans,n_last", [(False, 6), (True, 6), (True, 10)])
def test_affinity_weight_and_refine_vs_oracle(P, seg_trans, n_last):
    B, h, w = 2, 5, 7
    L = h * w + 1
    maps = _maps(B, L)
    g = torch.Generator().manual_seed(5)
    seg = torch.sigmoid(torch.randn(B, L - 1, L - 1, generator=g))
    Wd = P.affinity_weight([m.cuda() for m in maps], seg.cuda(), seg_trans, n_last)
    cams = torch.rand(3, h * w, generator=g)
    cams = (cams - cams.min(1, keepdim=True)[0]) / (cams.max(1, keepdim=True)[0] - cams.min(1, keepdim=True)[0])
    pair_img = torch.tensor([0, 0, 1], dtype=I32)
    pair_slot = torch.tensor([0, 1, 0], dtype=I32)
    R = P.refine(Wd, cams.cuda(), pair_img.cuda(), pair_slot.cuda(), 2, h, w, 0.4).cpu()
    for b in range(B):
        m12 = torch.stack([m[b] for m in maps])
        Wo = O.affinity_weight(m12, seg[b], seg_trans, n_last)
        np.testing.assert_allclose(Wd[b].cpu().numpy(), Wo.numpy(), rtol=2e-5, atol=1e-8)
        T = O.compute_trans_mat(Wo)
        for p in range(3):
            if pair_img[p] != b:
                continue
            cam = cams[p].reshape(h, w).numpy()
            ref = O.refine_cam(T, cam, O.box_mask(cam, 0.4)).reshape(-1)
            np.testing.assert_allclose(R[b, :, pair_slot[p]].numpy(), ref.numpy(), rtol=1e-4, atol=1e-8)


@pytest.mark.parametrize("thr", [0.4, 0.7])
def test_box_mask_matches_oracle_on_structured_cams(P, thr):
    """Blobs, diagonal (8-connectivity) links, border-touching boxes, empty and full maps."""
    h, w = 20, 28
    g = torch.Generator().manual_seed(3)
    cams = []
    for k in range(12):
        c = torch.rand(h, w, generator=g) * 0.3
        for _ in range(k % 5):
            y, x = int(torch.randint(0, h, (1,), generator=g)), int(torch.randint(0, w, (1,), generator=g))
            c[max(y - 2, 0):y + 2, max(x - 3, 0):x + 3] = 0.6 + 0.4 * torch.rand(1, generator=g)
        cams.append(c)
    d = torch.zeros(h, w); d[torch.arange(10), torch.arange(10)] = 1.0; cams.append(d)      # diagonal chain
    cams.append(torch.zeros(h, w)); cams.append(torch.ones(h, w))
    e = torch.zeros(h, w); e[h - 1, w - 1] = 1; e[0, 0] = 0.9; cams.append(e)               # corner pixels
    cams = torch.stack(cams).reshape(len(cams), -1)
    n = cams.shape[0]
    V, mask, boxes, nbox = P.box_masks(cams.cuda(), torch.zeros(n, dtype=I32).cuda(),
                                       torch.arange(n, dtype=I32).cuda(), 1, n, h, w, thr, True, True)
    for p in range(n):
        ref = O.box_mask(cams[p].reshape(h, w).numpy(), thr)
        assert (mask[p].cpu().numpy().reshape(h, w) == ref).all(), f"mask {p} differs"
        np.testing.assert_array_equal(V[0, :, p].cpu().numpy(), (ref * cams[p].reshape(h, w).numpy()).reshape(-1))


def test_box_mask_kernel_matches_hand_derived_opencv_semantics(P):
    """box_mask_kernel on the hand-derived vectors of tests/cv2_cases.py (documented OpenCV semantics, worked out by hand from
    clip/utils.py:115-142): strict threshold at int(thr * max), 8-connectivity, holes, the dropped border row / column,
    the empty map -> [[0,0,0,0]]."""
    from tests.cv2_cases import cases
    cs = cases()
    for thr in sorted({c[2] for c in cs}):
        sel = [c for c in cs if c[2] == thr]
        h, w = sel[0][1].shape
        cams = torch.stack([torch.from_numpy(c[1]).reshape(-1) for c in sel])
        n = len(sel)
        V, mask, boxes, nbox = P.box_masks(cams.cuda(), torch.zeros(n, dtype=I32).cuda(), torch.arange(n, dtype=I32).cuda(), 1, n, h, w,
                                           thr, True, True)
        for p, (name, cam, _, want) in enumerate(sel):
            got = mask[p].cpu().numpy().reshape(h, w)
            assert np.array_equal(got, want), (name, got, want)


def test_upsample_with_bg_vs_oracle(P):
    B, h, w, K, H, W = 2, 4, 6, 3, 64, 96
    g = torch.Generator().manual_seed(9)
    R = torch.rand(B, h * w, K, generator=g) * 0.01
    nk = torch.tensor([3, 2], dtype=I32)
    cams = P.upsample_with_bg(R.cuda(), nk.cuda(), h, w, H, W).cpu()
    for b in range(B):
        ups = torch.stack([O.upsample_cam(R[b, :, k].reshape(h, w), H, W) for k in range(int(nk[b]))])
        np.testing.assert_allclose(cams[b, 1:1 + int(nk[b])].numpy(), ups.numpy(), rtol=0, atol=2e-6)
        np.testing.assert_allclose(cams[b, 0].numpy(), (1 - ups.max(0)[0]).numpy(), rtol=0, atol=2e-6)
        assert (cams[b, 1 + int(nk[b]):] == 0).all()


def test_sinkhorn_properties_full_size(P):
    """hw = 1024 (512x512): T = diag(r) W diag(c) has unit row sums; T_sym is symmetric."""
    B, L = 2, 1025
    maps = _maps(B, L, n=12, seed=2)
    W = P.affinity_weight([m.cuda() for m in maps])
    r, c = P.sinkhorn_scales(W)
    T = r[:, :, None] * W * c[:, None, :]
    assert (T.sum(2) - 1).abs().max().item() < 1e-5
    assert (T.sum(1) - 1).abs().max().item() < 2e-2       # columns only approximately after 3 rounds
    X = torch.rand(B, L - 1, 2, device="cuda")
    Y = P.tsym_apply(W, r, c, X)
    Ts = 0.5 * (T + T.transpose(1, 2))
    assert (Y - Ts @ X).abs().max().item() < 1e-5


@pytest.mark.parametrize("h,w,K,seg_trans", [(6, 8, 2, False), (4, 12, 5, True), (32, 40, 3, False)])
def test_fused_sweeps_vs_oracle(P, h, w, K, seg_trans):
    """hw % 4 == 0 takes the fused sweeps (aff_weight + first column scale, Sinkhorn row+column passes in one read of W,
    both halves of T_sym X in one read; K > 4 classes in chunks of 4): W, the scale vectors and T_sym^2 V vs the oracle."""
    B = 3
    L = h * w + 1
    assert P.fused_ok(h * w)
    maps = _maps(B, L, n=12, seed=h + w)
    g = torch.Generator().manual_seed(h * w)
    seg = torch.sigmoid(torch.randn(B, L - 1, L - 1, generator=g))
    Wd, c1 = P.affinity_weight([m.cuda() for m in maps], seg.cuda(), seg_trans, 6, return_c1=True)
    assert c1 is not None
    V = torch.rand(B, h * w, K, generator=g)
    r, c = P.sinkhorn_scales(Wd, c1=c1)
    Y = P.tsym_apply(Wd, r, c, P.tsym_apply(Wd, r, c, V.cuda())).cpu()
    for b in range(B):
        Wo = O.affinity_weight(torch.stack([m[b] for m in maps]), seg[b], seg_trans, 6)
        np.testing.assert_allclose(Wd[b].cpu().numpy(), Wo.numpy(), rtol=2e-5, atol=1e-8)
        T = O.compute_trans_mat(Wo)                    # = T_sym @ T_sym
        np.testing.assert_allclose(Y[b].numpy(), (T @ V[b]).numpy(), rtol=2e-4, atol=1e-7)


def test_fused_and_unfused_paths_agree_at_full_size(P, monkeypatch):
    B, L = 2, 1025
    maps = [m.cuda() for m in _maps(B, L, n=12, seed=4)]
    X = torch.rand(B, L - 1, 2, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    W, c1 = P.affinity_weight(maps, return_c1=True)
    r, c = P.sinkhorn_scales(W, c1=c1)
    Y = P.tsym_apply(W, r, c, X)
    monkeypatch.setattr(P, "fused_ok", lambda hw: False)
    W2 = P.affinity_weight(maps)
    r2, c2 = P.sinkhorn_scales(W2)
    Y2 = P.tsym_apply(W2, r2, c2, X)
    assert torch.equal(W, W2)
    assert ((r - r2).abs() / r2.abs()).max().item() < 1e-5 and ((c - c2).abs() / c2.abs()).max().item() < 1e-5
    assert (Y - Y2).abs().max().item() < 1e-6 * max(Y2.abs().max().item(), 1.0) + 1e-7
