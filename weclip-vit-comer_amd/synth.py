"""Seeded synthetic weights / inputs for the WeCLIP hot path (used by bench.py, the parity tests and the
golden-fixture generator; re-exported as oracle.synth for the test infrastructure).

No pretrained CLIP checkpoint exists offline, so every parity test, the
golden fixtures and bench.py use weights drawn here.  The distributions follow
`CLIP.initialize_parameters` (reference clip/model.py:340-373): in_proj std = width^-0.5,
out_proj / c_proj std = width^-0.5 * (2*layers)^-0.5, c_fc std = (2*width)^-0.5, and the
values the reference keeps in fp16 (`convert_weights`, clip/model.py:457-478: Conv/Linear
weights+biases, `visual.proj`, `text_projection`) are rounded through fp16 so that a CPU
`model.float()` copy and our device copy hold bit-identical numbers.  `attn.in_proj_*`
stays full fp32 exactly like the reference (its isinstance test misses the myAtt fork).

This module imports nothing from /root/reference, from oracle/ or from the rest of the package (pure torch).
"""
import math

import numpy as np
import torch


# Tiny configuration shared by tests/golden/make_golden.py and the parity tests
# (width 64, 1 head, 12 vision blocks, 64x96 input, B=2, K=2).
TINY = dict(width=64, layers=12, embed_dim=32, text_width=64, gain=2.0, seed=0,
            logit_scale=math.log(25.0))
TINY_HW = (64, 96)
TINY_LABELS = [[3, 7], [0, 14]]
# Per-block CLS attention-sink strengths of the second benchmark-size seg-trans fixture (tests/golden/vitb_512_seg_sink.npz):
# the six layers the VOC seg-trans branch selects from get A_l = sum(map_l[1:, 1:]) between 987 and 1023, every layer >= 2.4
# away from their mean (the reference's fp32 sums have a quantum of 1/8 there), so the discrete selection carries signal.
SINK_512 = [0.0] * 6 + [3.0, 5.0, 1.0, 6.0, 4.0, 2.0]


def checksum(tensors):
    """Order-sensitive fp64 checksum used to detect RNG drift between fixture generation
    and test time."""
    s = 0.0
    for t in tensors:
        t = torch.as_tensor(t).double().flatten()
        s += float((t * torch.arange(1, t.numel() + 1, dtype=torch.float64).remainder(97)).sum())
    return np.float64(s)


def _h(t):
    """Round through fp16 (what `convert_weights` + `.float()` leaves on the CPU path)."""
    return t.half().float()


def make_clip_state_dict(width=768, layers=12, heads=None, patch=16, grid=14, embed_dim=512,
                         seed=0, gain=1.0, text_width=64, text_layers=1, vocab=49408,
                         with_text=True, logit_scale=math.log(100.0), qk_corr=0.8, cls_sink=None):
    """ViT-shaped CLIP state dict accepted by reference `build_model` (clip/model.py:481-529).

    `gain` scales the attention in-projection so softmax rows are not uniform; `qk_corr`
    correlates each block's key projection with its query projection so that similar tokens
    attend to each other (random independent Wq/Wk give structureless attention whose
    head/layer mean is uniform, which makes the affinity refinement degenerate).
    The text tower is a minimal stand-in (never on the hot path).

    `cls_sink` (one value per block, or None): makes the CLS token an attention sink of per-block strength, as in
    trained CLIP models.  Block l's query bias gets a component along the key the CLS token would have if its
    residual stream still held ln_pre(class_embedding + pos[0]), so every patch row's score against the CLS key rises
    by about cls_sink[l] and the mass A_l = sum(map_l[1:, 1:]) the seg-trans layer selection compares
    (clip/clip_tool.py:152-162) differs between blocks by far more than fp32 rounding.  Consumes no random draws: a
    state dict made with cls_sink=None is bit-identical to one made before the option existed.
    """
    g = torch.Generator().manual_seed(seed)

    def rn(*shape, std=1.0):
        return torch.randn(*shape, generator=g) * std

    sd = {}
    scale = width ** -0.5
    sd["visual.conv1.weight"] = _h(rn(width, 3, patch, patch, std=(3 * patch * patch) ** -0.5))
    sd["visual.class_embedding"] = rn(width, std=scale)
    sd["visual.positional_embedding"] = rn(grid * grid + 1, width, std=scale)
    for ln in ("visual.ln_pre", "visual.ln_post"):
        sd[ln + ".weight"] = 1.0 + rn(width, std=0.1)
        sd[ln + ".bias"] = rn(width, std=0.1)
    sd["visual.proj"] = _h(rn(width, embed_dim, std=scale))
    proj_std = (width ** -0.5) * ((2 * layers) ** -0.5)
    attn_std = width ** -0.5
    fc_std = (2 * width) ** -0.5
    for i in range(layers):
        p = f"visual.transformer.resblocks.{i}."
        w_in = rn(3 * width, width, std=attn_std * gain)
        w_in[width:2 * width] = qk_corr * w_in[:width] + math.sqrt(1 - qk_corr ** 2) * w_in[width:2 * width]
        sd[p + "attn.in_proj_weight"] = w_in
        sd[p + "attn.in_proj_bias"] = rn(3 * width, std=0.02)
        sd[p + "attn.out_proj.weight"] = _h(rn(width, width, std=proj_std))
        sd[p + "attn.out_proj.bias"] = _h(rn(width, std=0.02))
        sd[p + "ln_1.weight"] = 1.0 + rn(width, std=0.1)
        sd[p + "ln_1.bias"] = rn(width, std=0.1)
        sd[p + "mlp.c_fc.weight"] = _h(rn(4 * width, width, std=fc_std))
        sd[p + "mlp.c_fc.bias"] = _h(rn(4 * width, std=0.02))
        sd[p + "mlp.c_proj.weight"] = _h(rn(width, 4 * width, std=proj_std))
        sd[p + "mlp.c_proj.bias"] = _h(rn(width, std=0.02))
        sd[p + "ln_2.weight"] = 1.0 + rn(width, std=0.1)
        sd[p + "ln_2.bias"] = rn(width, std=0.1)
    if cls_sink is not None:
        assert len(cls_sink) == layers
        nh = heads or max(width // 64, 1)
        e0 = sd["visual.class_embedding"] + sd["visual.positional_embedding"][0]
        e0 = torch.nn.functional.layer_norm(e0, (width,), sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"], 1e-5)
        for i, beta in enumerate(cls_sink):
            p = f"visual.transformer.resblocks.{i}."
            a = torch.nn.functional.layer_norm(e0, (width,), sd[p + "ln_1.weight"], sd[p + "ln_1.bias"], 1e-5)
            k = sd[p + "attn.in_proj_weight"][width:2 * width] @ a + sd[p + "attn.in_proj_bias"][width:2 * width]
            kh = k.view(nh, -1)
            # score added per head ~ beta: (b_q . k_cls) / sqrt(dh) with b_q = beta * sqrt(dh) * k / |k|^2 per head
            bq = float(beta) * math.sqrt(kh.shape[1]) * kh / (kh * kh).sum(1, keepdim=True)
            sd[p + "attn.in_proj_bias"][:width] += bq.reshape(-1)
    sd["logit_scale"] = torch.tensor(float(logit_scale))
    if with_text:
        tw = text_width
        sd["positional_embedding"] = rn(77, tw, std=0.01)
        sd["text_projection"] = _h(rn(tw, embed_dim, std=tw ** -0.5))
        sd["token_embedding.weight"] = rn(vocab, tw, std=0.02)
        sd["ln_final.weight"] = torch.ones(tw)
        sd["ln_final.bias"] = torch.zeros(tw)
        for i in range(text_layers):
            p = f"transformer.resblocks.{i}."
            sd[p + "attn.in_proj_weight"] = rn(3 * tw, tw, std=tw ** -0.5)
            sd[p + "attn.in_proj_bias"] = torch.zeros(3 * tw)
            sd[p + "attn.out_proj.weight"] = _h(rn(tw, tw, std=tw ** -0.5))
            sd[p + "attn.out_proj.bias"] = torch.zeros(tw)
            sd[p + "ln_1.weight"] = torch.ones(tw)
            sd[p + "ln_1.bias"] = torch.zeros(tw)
            sd[p + "mlp.c_fc.weight"] = _h(rn(4 * tw, tw, std=(2 * tw) ** -0.5))
            sd[p + "mlp.c_fc.bias"] = torch.zeros(4 * tw)
            sd[p + "mlp.c_proj.weight"] = _h(rn(tw, 4 * tw, std=tw ** -0.5))
            sd[p + "mlp.c_proj.bias"] = torch.zeros(tw)
            sd[p + "ln_2.weight"] = torch.ones(tw)
            sd[p + "ln_2.bias"] = torch.zeros(tw)
    return sd


def make_text_features(n_fg=20, n_bg=25, embed_dim=512, seed=1, spread=0.35):
    """Unit-norm stand-ins for the zero-shot text classifier rows (reference
    WeCLIP_model/model_attn_aff_voc.py:34-46,81-82).  Like real CLIP prompt embeddings they
    sit in a narrow cone (common direction + `spread` x noise): with logit_scale = 100 this
    gives class probabilities of 1e-3..0.5 instead of ~1e-8, so the GradCAM gradient
    survives the reference's fp16 out-projection backward (clip/myAtt.py:321)."""
    g = torch.Generator().manual_seed(seed)
    common = torch.randn(1, embed_dim, generator=g)
    common = common / common.norm()
    noise = torch.randn(n_fg + n_bg, embed_dim, generator=g) / math.sqrt(embed_dim)
    t = common + spread * noise
    t = t / t.norm(dim=-1, keepdim=True)
    return t[n_fg:].contiguous(), t[:n_fg].contiguous()


def make_head_state_dicts(width=768, embedding_dim=256, num_classes=21, index=11, dec_layers=3,
                          seed=2):
    """Adapter (`SegFormerHead`) and decoder (`DecoderTransformer`) parameters with the
    reference key names (segformer_head.py:53-66, TransDecoder.py:104-110)."""
    g = torch.Generator().manual_seed(seed)

    def rn(*shape, std=1.0):
        return torch.randn(*shape, generator=g) * std

    E = embedding_dim
    fuse = {}
    for i in range(index):
        p = f"linears_modulelist.{i}."
        fuse[p + "proj.weight"] = rn(E, width, std=width ** -0.5)
        fuse[p + "proj.bias"] = rn(E, std=0.02)
        fuse[p + "proj_2.weight"] = rn(E, E, std=E ** -0.5)
        fuse[p + "proj_2.bias"] = rn(E, std=0.02)
    fuse["linear_fuse.weight"] = rn(E, E * index, 1, 1, std=(E * index) ** -0.5)
    fuse["linear_fuse.bias"] = rn(E, std=0.02)
    dec = {}
    for i in range(dec_layers):
        p = f"transformer.resblocks.{i}."
        dec[p + "attn.in_proj_weight"] = rn(3 * E, E, std=E ** -0.5)
        dec[p + "attn.in_proj_bias"] = rn(3 * E, std=0.02)
        dec[p + "attn.out_proj.weight"] = rn(E, E, std=E ** -0.5 * (2 * dec_layers) ** -0.5)
        dec[p + "attn.out_proj.bias"] = rn(E, std=0.02)
        dec[p + "ln_1.weight"] = 1.0 + rn(E, std=0.1)
        dec[p + "ln_1.bias"] = rn(E, std=0.1)
        dec[p + "mlp.c_fc.weight"] = rn(4 * E, E, std=(2 * E) ** -0.5)
        dec[p + "mlp.c_fc.bias"] = rn(4 * E, std=0.02)
        dec[p + "mlp.c_proj.weight"] = rn(E, 4 * E, std=E ** -0.5 * (2 * dec_layers) ** -0.5)
        dec[p + "mlp.c_proj.bias"] = rn(E, std=0.02)
        dec[p + "ln_2.weight"] = 1.0 + rn(E, std=0.1)
        dec[p + "ln_2.bias"] = rn(E, std=0.1)
    dec["linear_pred.weight"] = rn(num_classes, E, 1, 1, std=E ** -0.5)
    dec["linear_pred.bias"] = rn(num_classes, std=0.02)
    return fuse, dec


def make_images(batch, H, W, seed=100):
    """Mean/std-normalised pixels are ~N(0,1) (datasets/transforms.py:8-15).  A smooth
    low-frequency component is mixed in so PAR affinities and CAMs are structured."""
    g = torch.Generator().manual_seed(seed)
    noise = torch.randn(batch, 3, H, W, generator=g)
    coarse = torch.randn(batch, 3, max(H // 32, 1), max(W // 32, 1), generator=g)
    smooth = torch.nn.functional.interpolate(coarse, size=(H, W), mode="bilinear",
                                             align_corners=False)
    return (0.5 * noise + 1.0 * smooth).contiguous()


def make_label_lists(batch, k=2, n_classes=20, seed=7):
    """K distinct foreground class ids (0-based, sorted like np.unique) per image."""
    rs = np.random.RandomState(seed)
    return [sorted(rs.choice(n_classes, size=k, replace=False).tolist()) for _ in range(batch)]
