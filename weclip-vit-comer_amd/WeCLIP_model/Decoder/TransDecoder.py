"""Decoder: same module tree / keys as reference WeCLIP_model/Decoder/TransDecoder.py:63-125
(`transformer.resblocks.{i}.{attn.in_proj_weight,attn.in_proj_bias,attn.out_proj,ln_1,ln_2,
mlp.c_fc,mlp.c_proj}`, `linear_pred`).  3 pre-LN blocks (width 256, 8 heads) with the myAtt
quirks -- fp32 in-projection/softmax, out-projection forced to fp16 (clip/myAtt.py:199-201,321) --
then a 1x1 conv.  Trainable.  On the GPU the training step runs the three blocks, `linear_pred` and attn_pred forward and
backward inside head_engine.HeadEngine (HIP flash attention forward / backward, MFMA GEMMs, fused LayerNorm backward); these
modules own the parameters / state-dict keys, and their torch `forward` is the CPU path and the `WECLIP_HEAD=torch` A/B path.
"""
import math
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F


class LayerNorm(nn.LayerNorm):
    def forward(self, x):
        return super().forward(x.float()).type(x.dtype)


class QuickGELU(nn.Module):
    def forward(self, x):
        return x * torch.sigmoid(1.702 * x)


class _Attention(nn.Module):
    """Parameter container with the myAtt.MultiheadAttention names; differentiable forward."""

    def __init__(self, embed_dim, num_heads):
        super().__init__()
        self.embed_dim, self.num_heads, self.head_dim = embed_dim, num_heads, embed_dim // num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)

    def forward(self, x, need_weights=False):
        Lq, N, E = x.shape
        H, d = self.num_heads, self.head_dim
        qkv = F.linear(x.float(), self.in_proj_weight, self.in_proj_bias)
        q, k, v = qkv.chunk(3, dim=-1)
        q = q.contiguous().view(Lq, N * H, d).transpose(0, 1) / math.sqrt(d)
        k = k.contiguous().view(Lq, N * H, d).transpose(0, 1)
        v = v.contiguous().view(Lq, N * H, d).transpose(0, 1)
        p = torch.softmax(torch.bmm(q, k.transpose(1, 2)), dim=-1)
        o = torch.bmm(p, v).transpose(0, 1).contiguous().view(Lq, N, E)
        o = F.linear(o.half(), self.out_proj.weight.half(), self.out_proj.bias.half())
        w = p.view(N, H, Lq, Lq).sum(1) / H if need_weights else None
        return o, w


class ResidualAttentionBlock(nn.Module):
    fp32_mlp = True

    def __init__(self, d_model, n_head, attn_mask=None):
        super().__init__()
        self.attn = _Attention(d_model, n_head)
        self.ln_1 = LayerNorm(d_model)
        self.mlp = nn.Sequential(OrderedDict([
            ("c_fc", nn.Linear(d_model, d_model * 4)),
            ("gelu", QuickGELU()),
            ("c_proj", nn.Linear(d_model * 4, d_model)),
        ]))
        self.ln_2 = LayerNorm(d_model)

    def forward(self, x, need_weights=False):
        o, w = self.attn(self.ln_1(x), need_weights)
        x = x + o
        return x + self.mlp(self.ln_2(x)), w


class Transformer(nn.Module):
    def __init__(self, width, layers, heads, attn_mask=None):
        super().__init__()
        self.width, self.layers = width, layers
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads) for _ in range(layers)])

    def forward(self, x, need_weights=False):
        ws = []
        for blk in self.resblocks:
            x, w = blk(x, need_weights)
            ws.append(w)
        return x, ws


class DecoderTransformer(nn.Module):
    def __init__(self, width, layers, heads, output_dim):
        super().__init__()
        self.transformer = Transformer(width, layers, heads)
        self.linear_pred = nn.Conv2d(width, output_dim, kernel_size=1)

    def forward(self, x, need_weights=True):
        """(b, c, h, w) -> (logit (b, nc, h, w), [per-block head-mean attention (b, hw, hw)])."""
        b, c, h, w = x.shape
        t = x.reshape(b, c, h * w).permute(2, 0, 1)
        t, ws = self.transformer(t, need_weights)
        # 1x1 conv as a token GEMM (MIOpen has no tuned 1x1 kernel for this shape and falls back to
        # a naive convolution that costs more than the whole decoder)
        logit = F.linear(t, self.linear_pred.weight.flatten(1), self.linear_pred.bias)   # (hw, b, nc)
        return logit.permute(1, 2, 0).reshape(b, -1, h, w), ws
