"""Pin the CPU oracle (oracle/weclip_oracle.py) against outputs of the UNMODIFIED reference
captured in tests/golden/*.npz by tests/golden/make_golden.py (CPU only, no GPU)."""
import numpy as np
import pytest
import torch

from oracle import synth
from oracle import weclip_oracle as O

TINY, (H, W), LABELS = synth.TINY, synth.TINY_HW, synth.TINY_LABELS


@pytest.fixture(scope="module")
def tiny():
    sd = synth.make_clip_state_dict(**TINY)
    img = synth.make_images(2, H, W)
    bg, fg = synth.make_text_features(20, 25, TINY["embed_dim"])
    fuse, dec = synth.make_head_state_dicts(width=TINY["width"])
    return sd, img, bg, fg, fuse, dec


def _check_inputs(g, sd, img):
    if synth.checksum(sd.values()) != g["weights_ck"] or synth.checksum([img]) != g["img_ck"]:
        pytest.skip("synthetic RNG stream differs from the one the fixture was generated with")


def test_encoder_and_gradcam_match_reference(tiny, golden):
    sd, img, bg, fg, _, _ = tiny
    g = golden("tiny_func.npz")
    _check_inputs(g, sd, img)
    xs, maps = O.encode_image(img, sd, heads=1)
    assert len(xs) == 11 and len(maps) == 11
    np.testing.assert_allclose(xs[-1].numpy(), g["fts_last"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(xs[0].numpy(), g["fts_first"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(torch.stack(maps).numpy(), g["attn"], rtol=0, atol=1e-6)
    n = 0
    for i, ids in enumerate(LABELS):
        text = torch.cat([fg[ids], bg], 0)
        for j in range(len(ids)):
            cam, probs, pm, _ = O.grad_cam(xs[-1][:, i:i + 1], text, j, sd, 1, H // 16, W // 16)
            np.testing.assert_allclose(probs.numpy()[0], g["probs"][n], rtol=1e-4, atol=1e-8)
            np.testing.assert_allclose(pm.numpy()[0], g["attn_last"][n], rtol=0, atol=1e-6)
            np.testing.assert_allclose(cam, g["cams"][n], rtol=0, atol=2e-4)
            assert g["cams"][n].max() > 0.5          # fixture has signal
            n += 1


def test_trans_mat_matches_reference(golden):
    g = golden("tiny_func.npz")
    out = O.compute_trans_mat(torch.from_numpy(g["trans_in"]))
    np.testing.assert_allclose(out.numpy(), g["trans_out"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(out.numpy(), out.numpy().T, rtol=1e-5, atol=1e-9)   # symmetric


def test_par_matches_reference(tiny, golden):
    _, img, *_ = tiny
    g = golden("tiny_func.npz")
    out = O.par(img[:1], torch.from_numpy(g["par_masks"]))
    np.testing.assert_allclose(out.numpy(), g["par_out"], rtol=0, atol=2e-5)
    # mass property: aff sums to 1.01 => 1.01^20 growth
    ratio = out.sum().item() / g["par_masks"].sum()
    assert abs(ratio - 1.01 ** 20) < 0.02
    img2 = synth.make_images(1, 37, 53, seed=5)
    out2 = O.par(img2, torch.from_numpy(g["par2_masks"]), num_iter=3)
    np.testing.assert_allclose(out2.numpy(), g["par2_out"], rtol=0, atol=1e-5)


def test_head_and_decoder_match_reference(tiny, golden):
    sd, img, _, _, fuse, dec = tiny
    g = golden("tiny_func.npz")
    xs, _ = O.encode_image(img, sd, heads=1)
    toks = torch.stack(xs, 0)[:, 1:].permute(0, 2, 3, 1).reshape(11, 2, -1, H // 16, W // 16)
    f = O.segformer_head(toks, fuse)
    np.testing.assert_allclose(f.numpy(), g["head_out"], rtol=0, atol=2e-5)
    seg, maps = O.decoder(f, dec)
    np.testing.assert_allclose(seg.numpy(), g["dec_out"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(maps[0].numpy(), g["dec_map0"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("seg_trans", [False, True])
def test_whole_forward_and_backward_match_reference(tiny, golden, seg_trans):
    sd, img, bg, fg, fuse, dec = tiny
    g = golden("tiny_voc_seg.npz" if seg_trans else "tiny_voc.npz")
    _check_inputs(g, sd, img)
    fuse = {k: v.clone().requires_grad_(True) for k, v in fuse.items()}
    dec = {k: v.clone().requires_grad_(True) for k, v in dec.items()}
    seg, labels, ap = O.weclip_forward(img, LABELS, sd, fuse, dec, bg, fg, heads=1,
                                       seg_trans=seg_trans)
    np.testing.assert_allclose(seg.detach().numpy(), g["seg"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(ap.detach().numpy(), g["attn_pred"], rtol=0, atol=1e-5)
    mism = (labels.numpy() != g["cam_labels"]).mean()
    assert mism <= 1e-3, f"{mism:.2%} label pixels differ from the reference"
    assert len(np.unique(g["cam_labels"])) >= 3
    loss, ls, la = O.train_losses(seg, torch.from_numpy(g["cam_labels"].astype(np.int64)), ap)
    assert abs(ls.item() - g["seg_loss"]) < 1e-4 and abs(la.item() - g["attn_loss"]) < 1e-5
    al = O.cams_to_affinity_label(torch.from_numpy(g["cam_labels"].astype(np.int64)),
                                  O.radius_mask(H // 16, W // 16))
    assert (al.numpy().astype(np.uint8) == g["aff_label"]).all()
    loss.backward()
    grads = {**{k: v.grad for k, v in dec.items()}, **{k: v.grad for k, v in fuse.items()}}
    for k in g.files:
        if k.startswith("grad:"):
            np.testing.assert_allclose(grads[k[5:]].numpy(), g[k], rtol=1e-3, atol=1e-6)
    names = [str(n) for n in g["grad_names"]]
    norms = np.array([float(grads[n].norm()) for n in names])
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=2e-3, atol=1e-7)


@pytest.mark.parametrize("seg_trans", [False, True])
def test_coco_train_forward_matches_reference(golden, seg_trans):
    """The oracle's COCO configuration (80 + 23 prompts, CAM threshold 0.7, last 10 maps in the seg-trans branch) against the
    reference COCO model's own train forward."""
    g = golden("tiny_coco_train_seg.npz" if seg_trans else "tiny_coco_train.npz")
    sd = synth.make_clip_state_dict(**synth.TINY)
    img = synth.make_images(2, H, W, seed=600)
    _check_inputs(g, sd, img)
    bg, fg = synth.make_text_features(80, 23, synth.TINY["embed_dim"], seed=5)
    fuse, dec = synth.make_head_state_dicts(width=synth.TINY["width"], num_classes=81, seed=3)
    with torch.no_grad():
        seg, labels, ap = O.weclip_forward(img, [[2, 41], [0, 17, 79]], sd, fuse, dec, bg, fg, heads=1, seg_trans=seg_trans,
                                           dataset="coco")
    np.testing.assert_allclose(seg.numpy(), g["seg"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(ap.numpy(), g["attn_pred"], rtol=0, atol=1e-5)
    assert (labels.numpy() != g["cam_labels"]).mean() <= 1e-3
    assert set(np.unique(g["cam_labels"])) >= {0, 1, 42}


def test_vitb_224_config0_matches_reference(golden):
    """BASELINE config 0: ViT-B/16-sized weights, 224x224, encode + GradCAM + affinity."""
    g = golden("vitb_224.npz")
    sd = synth.make_clip_state_dict(seed=0, with_text=False)
    img = synth.make_images(1, 224, 224, seed=100)
    if synth.checksum([sd[k] for k in sorted(sd) if k.startswith("visual")]) != g["weights_ck"]:
        pytest.skip("synthetic RNG stream differs from fixture")
    bg, fg = synth.make_text_features(20, 25, 512)
    xs, maps = O.encode_image(img, sd, heads=12)
    np.testing.assert_allclose(xs[-1][:, 0].numpy(), g["fts_last"], rtol=0, atol=2e-4)
    np.testing.assert_allclose(maps[10][0, ::16].numpy(), g["attn10_rows"], rtol=0, atol=1e-6)
    text = torch.cat([fg[[0, 1]], bg], 0)
    for j in range(2):
        cam, probs, pm, _ = O.grad_cam(xs[-1], text, j, sd, 12, 14, 14)
        np.testing.assert_allclose(probs.numpy(), g["probs"], rtol=1e-3, atol=1e-7)
        np.testing.assert_allclose(cam, g["cams"][j], rtol=0, atol=1e-3)
        if j == 0:
            m12 = torch.cat([torch.stack([m[0] for m in maps]), pm], 0)
            T = O.compute_trans_mat(O.affinity_weight(m12))
            np.testing.assert_allclose(T[::28].numpy(), g["trans_rows"], rtol=1e-4, atol=1e-9)
        r = O.refine_cam(T, cam, O.box_mask(cam, 0.4))
        np.testing.assert_allclose(r.numpy(), g["refined"][j], rtol=1e-3, atol=1e-6)


def test_vitb_512_seg_sink_matches_reference(golden):
    """The benchmark-size seg-trans fixture WITH layer-selection signal (synth.SINK_512: the six candidate layers differ by
    2.4 ... 26 in A_l): the oracle's whole forward on that one image against the reference's own -- selection, transition
    matrix, refined CAMs, PAR output, labels."""
    g = golden("vitb_512_seg_sink.npz")
    sd = synth.make_clip_state_dict(seed=0, with_text=False, cls_sink=synth.SINK_512)
    if synth.checksum([sd[k] for k in sorted(sd) if k.startswith("visual")]) != g["weights_ck"]:
        pytest.skip("synthetic RNG stream differs from fixture")
    i = int(g["img_index"])
    img = synth.make_images(16, 512, 512, seed=100)[i:i + 1].contiguous()
    ids = synth.make_label_lists(16, 2, seed=7)[i]
    assert ids == g["ids"].tolist()
    bg, fg = synth.make_text_features(20, 25, 512)
    fuse, dec = synth.make_head_state_dicts()
    A64 = g["A64"]
    assert (g["keep_ref"] == (A64 >= A64.mean())).all() and np.abs(A64 - A64.mean()).min() > 1.0
    with torch.no_grad():
        seg, labels, ap, aux = O.weclip_forward(img, [ids], sd, fuse, dec, bg, fg, heads=12, seg_trans=True, return_aux=True)
    a = aux[0]
    np.testing.assert_allclose(a["probs"][0], g["probs"][0], rtol=1e-3, atol=1e-7)
    np.testing.assert_allclose(a["cams"], g["cams"], rtol=0, atol=1e-3)
    np.testing.assert_allclose(a["trans"][::64], g["trans_rows"], rtol=2e-3, atol=1e-9)
    np.testing.assert_allclose(a["refined"], g["refined"], rtol=2e-3, atol=1e-6)
    np.testing.assert_allclose(a["par"][:, ::16], g["par_out_rows"], rtol=0, atol=1e-3)
    np.testing.assert_allclose(seg[0].numpy(), g["seg"], rtol=0, atol=2e-4)
    assert (labels[0].numpy() != g["cam_labels"]).mean() <= 5e-4


def _augment_inputs():
    f = synth.make_images(6, 54, 76, seed=700)
    return (f * 58.0 + 118.0).clamp_(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()


def test_input_pipeline_oracle_matches_reference_transforms(golden):
    """f-1: the oracle's restatement of the reference's train-time chain (datasets/transforms.py random_scaling with PIL
    BILINEAR, random_fliplr, random_crop, normalize_img; order of datasets/voc.py:109-143) against the fixture the
    reference's own functions produced -- every pixel EQUAL, up- and down-scaling -- and the host-side draws of
    `data.DeviceAugment` against the recorded draws of the reference's random sources."""
    from weclip_vit_comer_amd.data import DeviceAugment
    g = golden("augment_ref.npz")
    imgs = _augment_inputs()
    assert synth.checksum([imgs]) == g["img_ck"]
    crop, d = int(g["crop"]), g["draws"]
    assert (d[:, 0] < 0.9).sum() >= 2 and (d[:, 0] > 1.1).sum() >= 2
    aug = DeviceAugment(crop_size=crop, rescale_range=(0.5, 2.0), seed=int(g["seed"]))
    for b in range(len(imgs)):
        s, flip, rh, rw, pad_y, pad_x, crop_y, crop_x = aug.draw_one(54, 76)
        assert [s, pad_y, pad_x, crop_y, crop_x] == [d[b, 0], d[b, 2], d[b, 3], d[b, 4], d[b, 5]] and flip == int(d[b, 1] > 0.5)
        out = O.augment_normalize(imgs[b], s, flip, pad_y, pad_x, crop_y, crop_x, crop)
        assert np.array_equal(out.numpy(), g["out"][b]), (b, s, np.abs(out.numpy() - g["out"][b]).max())
        # img_box of random_crop (transforms.py:160-164) from the same draws
        box = [max(pad_y - crop_y, 0), min(crop_y + crop, pad_y + rh), max(pad_x - crop_x, 0), min(crop_x + crop, pad_x + rw)]
        assert box == g["img_box"][b].tolist()


def test_pil_bilinear_restatement_equals_pillow():
    """`O.pil_bilinear_u8` against the installed Pillow itself (same library the reference calls), random sizes and ratios."""
    Image = pytest.importorskip("PIL.Image")
    rs = np.random.RandomState(3)
    for t in range(12):
        H, W = int(rs.randint(8, 90)), int(rs.randint(8, 120))
        s = float(rs.uniform(0.3, 2.3))
        img = rs.randint(0, 256, (H, W, 3)).astype(np.uint8)
        rw, rh = max(int(s * W), 1), max(int(s * H), 1)
        ref = np.asarray(Image.fromarray(img).resize([rw, rh], resample=Image.BILINEAR))
        assert np.array_equal(O.pil_bilinear_u8(img, rw, rh), ref), (H, W, s)


def test_vitb_512_train_losses_and_gradients_match_reference(golden):
    """The oracle's trainable half at the benchmark resolution (adapters, decoder, attn_pred, losses, backward) against the
    reference's own `loss.backward()` on image 3 of the benchmark batch."""
    g = golden("vitb_512_train.npz")
    sd = synth.make_clip_state_dict(seed=0, with_text=False)
    if synth.checksum([sd[k] for k in sorted(sd) if k.startswith("visual")]) != g["weights_ck"]:
        pytest.skip("synthetic RNG stream differs from fixture")
    i = int(g["img_index"])
    img = synth.make_images(16, 512, 512, seed=100)[i:i + 1].contiguous()
    bg, fg = synth.make_text_features(20, 25, 512)
    fuse, dec = synth.make_head_state_dicts()
    fuse = {k: v.requires_grad_(True) for k, v in fuse.items()}
    dec = {k: v.requires_grad_(True) for k, v in dec.items()}
    seg, labels, ap = O.weclip_forward(img, [g["ids"].tolist()], sd, fuse, dec, bg, fg, heads=12)
    assert (labels[0].numpy() != g["cam_labels"]).mean() <= 5e-4
    ref_labels = torch.from_numpy(g["cam_labels"].astype(np.int64))[None]
    loss, seg_loss, attn_loss = O.train_losses(seg, ref_labels, ap)
    assert abs(seg_loss.item() - float(g["seg_loss"])) < 1e-4 and abs(attn_loss.item() - float(g["attn_loss"])) < 1e-5
    loss.backward()
    grads = {**{k: v.grad for k, v in dec.items()}, **{k: v.grad for k, v in fuse.items()}}
    names = [str(n) for n in g["grad_names"]]
    np.testing.assert_allclose(np.array([float(grads[n].norm()) for n in names]), g["grad_norms"], rtol=2e-3, atol=1e-7)
    for k in g.files:
        if k.startswith("grad:"):
            np.testing.assert_allclose(grads[k[5:]].numpy().reshape(g[k].shape), g[k], rtol=2e-3, atol=1e-6 * np.abs(g[k]).max() + 1e-9)


def test_vitb_320_reference_default_crop(golden):
    """The oracle's whole forward at the reference's own default geometry (320 x 320) against the reference's."""
    g = golden("vitb_320.npz")
    sd = synth.make_clip_state_dict(seed=0, with_text=False)
    if synth.checksum([sd[k] for k in sorted(sd) if k.startswith("visual")]) != g["weights_ck"]:
        pytest.skip("synthetic RNG stream differs from fixture")
    i, S3, B3 = int(g["img_index"]), int(g["size"]), int(g["batch"])
    img = synth.make_images(B3, S3, S3, seed=int(g["seed"]))[i:i + 1].contiguous()
    ids = synth.make_label_lists(B3, 2, seed=int(g["label_seed"]))[i]
    assert ids == g["ids"].tolist()
    bg, fg = synth.make_text_features(20, 25, 512)
    fuse, dec = synth.make_head_state_dicts()
    with torch.no_grad():
        seg, labels, ap, aux = O.weclip_forward(img, [ids], sd, fuse, dec, bg, fg, heads=12, return_aux=True)
    np.testing.assert_allclose(aux[0]["probs"][0], g["probs"][0], rtol=1e-3, atol=1e-7)
    np.testing.assert_allclose(aux[0]["refined"], g["refined"], rtol=2e-3, atol=1e-6)
    np.testing.assert_allclose(seg[0].numpy(), g["seg"], rtol=0, atol=2e-4)
    assert (labels[0].numpy() != g["cam_labels"]).mean() <= 5e-4


def test_box_step_matches_hand_derived_opencv_semantics():
    """The oracle's scoremap2bbox + fill against hand-derived vectors (tests/cv2_cases.py: every case cites the OpenCV
    documentation it follows).  Keeps "unverified vs real cv2" (no OpenCV in the image) but pins the documented semantics."""
    from tests.cv2_cases import cases
    for name, cam, thr, want in cases():
        got = O.box_mask(cam, thr)
        assert np.array_equal(got, want), (name, got, want)


def test_cv2_stand_in_of_the_fixture_generator_matches_hand_derived_vectors():
    """The functional cv2 stand-in that the fixture generator installs under the imported reference (oracle/refharness.py)
    run through the reference's own sequence of calls (clip/utils.py:115-142, clip_tool.py:179-183 restated on the stub's
    functions): same hand-derived masks, so the committed fixtures were produced under the documented semantics too."""
    from oracle import refharness
    from tests.cv2_cases import cases
    cv2 = refharness._cv2_stub()
    for name, cam, thr, want in cases():
        h, w = cam.shape
        img = np.expand_dims((cam * 255).astype(np.uint8), 2)
        _, binm = cv2.threshold(src=img, thresh=int(thr * np.max(img)), maxval=255, type=cv2.THRESH_BINARY)
        contours = cv2.findContours(image=binm, mode=cv2.RETR_TREE, method=cv2.CHAIN_APPROX_SIMPLE)[0]
        boxes = [[0, 0, 0, 0]]
        if len(contours):
            boxes = []
            for c in contours:
                x, y, bw, bh = cv2.boundingRect(c)
                boxes.append([x, y, min(x + bw, w - 1), min(y + bh, h - 1)])
        got = np.zeros((h, w), np.float32)
        for x0, y0, x1, y1 in boxes:
            got[y0:y1, x0:x1] = 1
        assert np.array_equal(got, want), (name, got, want)
