"""Frozen CLIP ViT encoder (K1-K6) on the HIP path vs the reference goldens / CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import synth
from oracle import weclip_oracle as O

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return ((a - b).abs().max() / b.abs().max()).item()


@pytest.fixture(scope="module")
def clipmod():
    from weclip_vit_comer_amd import clip
    return clip


@pytest.mark.parametrize("precision,tol_x,tol_map", [("fast", 3e-3, 3e-3), ("exact", 1e-3, 3e-3)])
def test_tiny_encoder_matches_reference_golden(clipmod, golden, precision, tol_x, tol_map):
    from weclip_vit_comer_amd import config
    config.precision = precision
    try:
        g = golden("tiny_func.npz")
        sd = synth.make_clip_state_dict(**synth.TINY)
        H, W = synth.TINY_HW
        img = synth.make_images(2, H, W)
        model, _ = clipmod.load(sd, device="cuda")
        fts, attns = model.encode_image(img.cuda(), H, W, require_all_fts=True)
        assert len(fts) == 11 and len(attns) == 11
        assert tuple(fts[0].shape) == (25, 2, 64) and tuple(attns[0].shape) == (2, 25, 25)
        e0, e1 = _rel(fts[0].cpu(), g["fts_first"]), _rel(fts[-1].cpu(), g["fts_last"])
        em = _rel(torch.stack(attns).cpu(), g["attn"])
        print(f"[{precision}] tiny: block1 {e0:.2e} block11 {e1:.2e} maps {em:.2e}")
        assert e0 < tol_x and e1 < tol_x and em < tol_map
    finally:
        config.precision = "fast"


def test_state_dict_keys_match_reference_names(clipmod):
    sd = synth.make_clip_state_dict(**synth.TINY)
    model, _ = clipmod.load(sd, device="cuda")
    keys = set(model.state_dict().keys())
    assert set(sd.keys()) <= keys
    blk = model.visual.transformer.resblocks[-1]
    assert hasattr(blk, "ln_1") and hasattr(blk.attn, "in_proj_weight")


@pytest.mark.parametrize("precision,tol", [("fast", 2e-3), ("exact", 1e-3)])
def test_vitb_224_matches_reference_golden(clipmod, golden, precision, tol):
    from weclip_vit_comer_amd import config
    config.precision = precision
    try:
        g = golden("vitb_224.npz")
        sd = synth.make_clip_state_dict(seed=0, with_text=False)
        img = synth.make_images(1, 224, 224, seed=100)
        model, _ = clipmod.load(sd, device="cuda")
        fts, attns = model.encode_image(img.cuda(), 224, 224, require_all_fts=True)
        e_x = _rel(fts[-1][:, 0].cpu(), g["fts_last"])
        e_m = _rel(attns[10][0, ::16].cpu(), g["attn10_rows"])
        print(f"[{precision}] vitb224 feature rel err {e_x:.2e}, map rel err {e_m:.2e}")
        assert e_x < tol and e_m < 10 * tol
    finally:
        config.precision = "fast"


def test_encoder_512_properties(clipmod):
    """BASELINE size: 2 x 512x512 (L = 1025): attention rows sum to 1, finite, oracle agreement."""
    sd = synth.make_clip_state_dict(seed=0, with_text=False)
    img = synth.make_images(2, 512, 512, seed=101)
    model, _ = clipmod.load(sd, device="cuda")
    fts, attns = model.encode_image(img.cuda(), 512, 512, require_all_fts=True)
    assert tuple(fts[-1].shape) == (1025, 2, 768) and tuple(attns[-1].shape) == (2, 1025, 1025)
    for a in attns:
        assert (a.sum(-1) - 1).abs().max().item() < 2e-4
    assert all(torch.isfinite(f).all() for f in fts)
    xs, maps = O.encode_image(img[:1], sd, 12)
    ex, em = _rel(fts[-1][:, 0].cpu(), xs[-1][:, 0]), _rel(attns[-1][0].cpu(), maps[-1][0])
    print(f"[fast] 512: feature rel err {ex:.2e}, map rel err {em:.2e}")
    assert ex < 2e-3 and em < 2e-2
