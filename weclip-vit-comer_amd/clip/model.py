"""CLIP ViT visual tower on the HIP path, with the reference's module tree and parameter names
(clip/model.py:177-287,297-429) so `state_dict()` keys match and reference checkpoints load.

Differences from a plain CLIP that the reference introduced and this keeps:
  * position embedding bilinearly resized to any H/16 x W/16 grid and rounded through fp16;
  * `encode_image(..., require_all_fts=True)` runs blocks 1..layers-1 and returns the list of all
    token tensors (L,B,D) and all head-averaged attention maps (B,L,L);
  * `forward_last_layer` = last block + ln_post + patch-mean + proj + cosine softmax.
The text tower is not on the hot path: its parameters are kept for
checkpoint compatibility, `encode_text` is not provided.
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from .. import _lib as L
from .. import config, ops
from ..ops import F16, F32, Split
from ..resize import bilinear_resize
from . import myAtt
from . import vit_engine as VE


class LayerNorm(nn.LayerNorm):
    """fp32 LayerNorm; forward hooks work (GradCAM-style tooling hooks `resblocks[-1].ln_1`)."""

    def forward(self, x):
        shp = x.shape
        y32, _ = ops.layernorm(x.detach().float().contiguous().view(-1, shp[-1]),
                               self.weight.detach().float(), self.bias.detach().float(),
                               eps=self.eps, want32=True, want16=False)
        return y32.view(shp).to(x.dtype)


class QuickGELU(nn.Module):
    def forward(self, x):   # fused into the c_fc GEMM epilogue on the block path
        return x * torch.sigmoid(1.702 * x)


class ResidualAttentionBlock(nn.Module):
    fp32_mlp = False   # CLIP keeps c_fc/c_proj in fp16 (convert_weights): no lo-part needed

    def __init__(self, d_model, n_head, attn_mask=None):
        super().__init__()
        self.attn = myAtt.MultiheadAttention(d_model, n_head)
        self.ln_1 = LayerNorm(d_model)
        self.mlp = nn.Sequential(OrderedDict([
            ("c_fc", nn.Linear(d_model, d_model * 4)),
            ("gelu", QuickGELU()),
            ("c_proj", nn.Linear(d_model * 4, d_model)),
        ]))
        self.ln_2 = LayerNorm(d_model)
        self.attn_mask = attn_mask
        self._pack = None

    def pack(self, refresh=False):
        dev = self.ln_1.weight.device
        if refresh or self._pack is None or self._pack.ln1_w.device != dev or \
                self._pack.exact != config.exact():
            self._pack = VE.BlockPack(self)
        return self._pack

    def forward(self, x):
        """x (L, N, E) -> (x', head-mean attention (N, L, L))."""
        rows, N, Lq = VE.to_rows(x)
        y, mean = VE.run_block(self.pack(refresh=self.training or self.fp32_mlp), rows, N, Lq)
        return VE.from_rows(y, N, Lq).to(x.dtype), mean


class Transformer(nn.Module):
    def __init__(self, width, layers, heads, attn_mask=None):
        super().__init__()
        self.width, self.layers = width, layers
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads, attn_mask)
                                         for _ in range(layers)])

    def run_rows(self, rows, B, Lq, n_layers, want_maps=True):
        xs, maps = [], []
        for i in range(n_layers):
            rows, m = VE.run_block(self.resblocks[i].pack(), rows, B, Lq, want_mean=want_maps)
            xs.append(rows)
            maps.append(m)
        return xs, maps

    def forward(self, x, require_all_fts=False):
        # reference: all layers for the 77-token text input, layers-1 for vision (model.py:229)
        n = self.layers if x.shape[0] == 77 else self.layers - 1
        rows, N, Lq = VE.to_rows(x)
        xs, maps = self.run_rows(rows, N, Lq, n)
        outs = [VE.from_rows(r, N, Lq) for r in xs]
        return (outs, maps) if require_all_fts else (outs[-1], maps)


class VisionTransformer(nn.Module):
    def __init__(self, input_resolution, patch_size, width, layers, heads, output_dim):
        super().__init__()
        self.input_resolution, self.output_dim, self.patch_size = input_resolution, output_dim, patch_size
        self.conv1 = nn.Conv2d(3, width, kernel_size=patch_size, stride=patch_size, bias=False)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(
            scale * torch.randn((input_resolution // patch_size) ** 2 + 1, width))
        self.ln_pre = LayerNorm(width)
        self.transformer = Transformer(width, layers, heads)
        self.ln_post = LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))
        self._pos_cache = {}
        self._conv_w = None

    def resized_pos(self, h, w):
        """upsample_pos_emb (clip/model.py:11-27), cached per grid (the encoder is frozen)."""
        key = (h, w, self.positional_embedding.device, self.positional_embedding._version)
        if key not in self._pos_cache:
            pos = self.positional_embedding.detach().float()
            n, d = pos.shape[0] - 1, pos.shape[1]
            s = int(round(np.sqrt(n)))
            grid = pos[1:].t().reshape(1, d, s, s).contiguous()
            up = bilinear_resize(grid, (h, w), align_corners=False).reshape(d, h * w).t()
            full = torch.cat([pos[:1], up], 0).half().float().contiguous()   # fp16 round, model.py:26
            self._pos_cache = {key: full}
        return self._pos_cache[key]

    def embed(self, img):
        """img (B,3,H,W) -> ln_pre'd tokens (B*L, D) fp32 (clip/model.py:266-273)."""
        B, _, H, W = img.shape
        P = self.patch_size
        h, w = H // P, W // P
        hw, Lq = h * w, h * w + 1
        D = self.conv1.weight.shape[0]
        K = 3 * P * P
        dev = img.device
        ex = config.exact()
        a = Split(torch.empty(B * hw, K, device=dev, dtype=F16),
                  torch.empty(B * hw, K, device=dev, dtype=F16) if ex else None)
        img = img.detach().float().contiguous()
        L.lib().wc_patchify(L.ptr(img, F32, "img"), L.ptr(a.hi), L.ptr(a.lo), B, H, W, P, L.stream())
        if self._conv_w is None or self._conv_w.hi.device != dev:
            self._conv_w = ops.split_f16(self.conv1.weight.detach().reshape(D, K))
        pos = self.resized_pos(h, w)
        x = torch.empty(B, Lq, D, device=dev, dtype=F32)
        # rows 1.. of every image = patches @ W^T + pos[1:]  (batched: one image per z-slice)
        ops.gemm(a, self._conv_w, hw, D, K, out32=x.view(-1)[D:], ldc=D, resid=pos[1:], ldr=D, sR=0,
                 batch=B, sA=hw * K, sW=0, sC=Lq * D)
        L.lib().wc_cls_rows(L.ptr(x), L.ptr(self.class_embedding.detach().float(), F32),
                            L.ptr(pos, F32), B, Lq, D, L.stream())
        y, _ = ops.layernorm(x.view(B * Lq, D), self.ln_pre.weight.detach().float(),
                             self.ln_pre.bias.detach().float(), want32=True, want16=False)
        return y, B, Lq

    def forward(self, x, H, W, require_all_fts=False):
        rows, B, Lq = self.embed(x)
        tr = self.transformer
        xs, maps = tr.run_rows(rows, B, Lq, tr.layers - 1)
        outs = [VE.from_rows(r, B, Lq) for r in xs]
        return (outs, maps) if require_all_fts else (outs[-1], maps)


class CLIP(nn.Module):
    def __init__(self, embed_dim, image_resolution, vision_layers, vision_width, vision_patch_size,
                 context_length, vocab_size, transformer_width, transformer_heads, transformer_layers):
        super().__init__()
        if isinstance(vision_layers, (tuple, list)):
            raise NotImplementedError("ModifiedResNet visual towers are outside the WeCLIP hot path")
        self.context_length = context_length
        self.visual = VisionTransformer(image_resolution, vision_patch_size, vision_width, vision_layers,
                                        vision_width // 64, embed_dim)
        # text tower: parameters only (checkpoint compatibility); never executed here
        self.transformer = Transformer(transformer_width, transformer_layers, transformer_heads)
        self.vocab_size = vocab_size
        self.token_embedding = nn.Embedding(vocab_size, transformer_width)
        self.positional_embedding = nn.Parameter(torch.empty(context_length, transformer_width))
        self.ln_final = LayerNorm(transformer_width)
        self.text_projection = nn.Parameter(torch.empty(transformer_width, embed_dim))
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))

    @property
    def dtype(self):
        return self.visual.conv1.weight.dtype

    def encode_image(self, image, H, W, require_all_fts=False):
        return self.visual(image.type(self.dtype), H, W, require_all_fts=require_all_fts)

    def encode_text(self, text):
        raise NotImplementedError(
            "the text tower is init-time only and not on the HIP path: precompute text features "
            "(e.g. with the reference CLIP) and pass them to WeCLIP(text_features=...)")

    def forward_last_layer(self, image_features, text_features):
        """(L,N,D) tokens of block layers-1, (T,E) text rows -> (softmax probs (N,T), map (N,L,L)).
        clip/model.py:407-429.  Not differentiable: GradCAM uses the explicit analytic backward
        (pytorch_grad_cam.GradCAM of this package)."""
        from ..gradcam_engine import last_layer_forward
        rows, N, Lq = VE.to_rows(image_features)
        st = last_layer_forward(self, rows, N, Lq)
        probs = st.class_probs(text_features.detach().float().contiguous())
        return probs, st.mean


def convert_weights(model):
    """Round the tensors the reference keeps in fp16 (clip/model.py:457-478: Conv/Linear weights and
    biases, `proj`, `text_projection`) through fp16; storage stays fp32 on this path."""
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, (nn.Conv1d, nn.Conv2d, nn.Linear)):
                m.weight.copy_(m.weight.half().float())
                if m.bias is not None:
                    m.bias.copy_(m.bias.half().float())
            for name in ("text_projection", "proj"):
                t = getattr(m, name, None)
                if isinstance(t, torch.Tensor):
                    t.copy_(t.half().float())


def build_model(state_dict):
    """CLIP from a ViT state dict (shapes inferred like reference clip/model.py:481-529)."""
    sd = dict(state_dict)
    if "visual.proj" not in sd:
        raise NotImplementedError("only ViT CLIP checkpoints are supported on the WeCLIP hot path")
    vw = sd["visual.conv1.weight"].shape[0]
    vl = len([k for k in sd if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
    ps = sd["visual.conv1.weight"].shape[-1]
    grid = round((sd["visual.positional_embedding"].shape[0] - 1) ** 0.5)
    embed_dim = sd["visual.proj"].shape[1]
    has_text = "text_projection" in sd
    if has_text:
        ctx, vocab = sd["positional_embedding"].shape[0], sd["token_embedding.weight"].shape[0]
        tw = sd["ln_final.weight"].shape[0]
        tl = len({k.split(".")[2] for k in sd if k.startswith("transformer.resblocks")})
    else:   # vision-only dict: a placeholder text tower of minimal size
        ctx, vocab, tw, tl = 77, 1, 64, 0
    model = CLIP(embed_dim, ps * grid, vl, vw, ps, ctx, vocab, tw, max(tw // 64, 1), tl)
    for k in ("input_resolution", "context_length", "vocab_size"):
        sd.pop(k, None)
    sd = {k: v.float() for k, v in sd.items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    if unexpected or any(k.startswith("visual.") or k == "logit_scale" for k in missing):
        raise RuntimeError(f"CLIP state dict mismatch: missing {missing}, unexpected {unexpected}")
    convert_weights(model)
    return model.eval()
