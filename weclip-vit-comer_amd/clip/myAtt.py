"""`MultiheadAttention` with the reference fork's contract (clip/myAtt.py:333-495): self-attention
that always returns the head-averaged probabilities, fp32 in-projection / softmax and an
out-projection forced to fp16.  Parameters keep the reference names (`in_proj_weight`,
`in_proj_bias`, `out_proj.weight`, `out_proj.bias`) so state dicts are interchangeable.
The arithmetic is csrc/gemm.hip + csrc/attention.hip."""
import torch
import torch.nn as nn

from .. import config, ops
from ..ops import F16, F32
from . import vit_engine as VE


class MultiheadAttention(nn.Module):
    def __init__(self, embed_dim, num_heads):
        super().__init__()
        if embed_dim % num_heads:
            raise AssertionError(f"embed_dim {embed_dim} not divisible by num_heads {num_heads}")
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.head_dim = embed_dim // num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)

    def forward(self, query, key, value, need_weights=True, attn_mask=None):
        """query=key=value (L, N, E).  Returns (out (L,N,E) holding fp16-rounded values,
        weights (N,L,L) mean over heads).  Inference only (no autograd)."""
        if key is not query or value is not query:
            raise RuntimeError("only self-attention (query is key is value) is on the HIP path")
        if attn_mask is not None:
            raise RuntimeError("attn_mask is not supported on the vision path")
        x, N, Lq = VE.to_rows(query)
        E, H, DH = self.embed_dim, self.num_heads, self.head_dim
        ex = config.exact()
        a = ops.split_f16(x, with_lo=ex)
        w = ops.split_f16(self.in_proj_weight, with_lo=ex)
        qkv = torch.empty(N * Lq, 3 * E, device=x.device, dtype=F16)
        ops.gemm(a, w, N * Lq, 3 * E, E, bias=self.in_proj_bias.detach().float(), out16=qkv,
                 scale=ops.q_scale(DH), scale_cols=E)
        o16, _, mean = ops.attention(qkv, N, Lq, H, DH, want_mean=need_weights)
        out = torch.empty(N * Lq, E, device=x.device, dtype=F32)
        ops.gemm(o16, ops.split_f16(self.out_proj.weight), N * Lq, E, E,
                 bias=self.out_proj.bias.detach().float().half().float(), out32=out, round16=True)
        return VE.from_rows(out, N, Lq), mean
