"""Transformer-block execution on the HIP path (shared by the CLIP ViT and the WeCLIP decoder).

Activations are kept image-major: a token tensor is a dense fp32 matrix (B*L, E) whose row
b*L + l is token l of image b (the reference keeps (L, B, E); only the API boundary permutes).
One residual block (reference clip/model.py:210-214 + clip/myAtt.py:199-326) is 8 launches:
  LN1 -> fp16 | in-proj GEMM (+bias, q*log2e/sqrt(dh)) -> fp16 qkv | V^T | flash attention
  (+ head-mean map) | out-proj GEMM (fp16-rounded, + residual) | LN2 -> fp16 |
  c_fc GEMM (+bias, QuickGELU) -> fp16 | c_proj GEMM (+bias, + residual).
"""
import torch

from .. import config, ops
from ..ops import F16, F32, Split


class BlockPack:
    """fp16 MFMA operands + fp32 vectors of one residual attention block."""

    def __init__(self, blk, exact=None, pre=None):
        """pre: optional dict of ready fp16 operands (in_w, out_w, fc_w, pj_w) -- the trainable decoder converts
        all its weights in one launch per step (ops.WeightCache)."""
        exact = config.exact() if exact is None else exact
        pre = pre or {}
        at = blk.attn
        self.E = at.embed_dim
        self.H = at.num_heads
        self.DH = self.E // self.H
        self.exact = exact
        f = lambda p: p.detach().float().contiguous()
        self.ln1_w, self.ln1_b = f(blk.ln_1.weight), f(blk.ln_1.bias)
        self.ln2_w, self.ln2_b = f(blk.ln_2.weight), f(blk.ln_2.bias)
        self.in_w = pre.get("in_w") or ops.split_f16(at.in_proj_weight, with_lo=exact)
        self.in_b = f(at.in_proj_bias)
        # reference forces this GEMM to fp16 on every device (myAtt.py:321)
        self.out_w = pre.get("out_w") or ops.split_f16(at.out_proj.weight, with_lo=False)
        self.out_b = f(at.out_proj.bias).half().float()
        self.fc_w = pre.get("fc_w") or ops.split_f16(blk.mlp.c_fc.weight, with_lo=exact and blk.fp32_mlp)
        self.fc_b = f(blk.mlp.c_fc.bias)
        self.pj_w = pre.get("pj_w") or ops.split_f16(blk.mlp.c_proj.weight, with_lo=exact and blk.fp32_mlp)
        self.pj_b = f(blk.mlp.c_proj.bias)


class X16Stack(list):
    """List of the fp16 block outputs that also owns ONE (n, M, E) buffer they are slices of, so that the adapter
    GEMMs over all blocks can run as a grouped launch (uniform stride between blocks): `big` / `big_lo`."""

    def __init__(self, n):
        super().__init__()
        self.n, self.big, self.big_lo = n, None, None

    def next_slot(self, M, E, dev, ex):
        if self.big is None:
            self.big = torch.empty(self.n, M, E, device=dev, dtype=F16)
            self.big_lo = torch.empty(self.n, M, E, device=dev, dtype=F16) if ex else None
        i = len(self)
        if i >= self.n or self.big.shape[1:] != (M, E):
            return None
        return Split(self.big[i], self.big_lo[i] if self.big_lo is not None else None)


def run_block(pk, x, B, L, want_mean=True, keep=None, x16_out=None, tag=None):
    """x (B*L, E) fp32 -> (x_out fp32, head-mean map (B,L,L) or None).
    `keep`, if a dict, receives intermediates needed by the analytic backward.
    `tag`: bench.py's roofline group of the attention half (in-projection, attention, head-mean, out-projection)."""
    M, E, H, DH = B * L, pk.E, pk.H, pk.DH
    dev = x.device
    ex = pk.exact
    a32, a = ops.layernorm(x, pk.ln1_w, pk.ln1_b, want32=keep is not None, with_lo=ex)
    if tag:
        ops.KernelTimer.tag(tag)
    qkv = torch.empty(M, 3 * E, device=dev, dtype=F16)
    ops.gemm(a, pk.in_w, M, 3 * E, E, bias=pk.in_b, out16=qkv, scale=ops.q_scale(DH), scale_cols=E)
    o32 = None
    if keep is not None:
        o16, lse, mean, o32 = ops.attention(qkv, B, L, H, DH, want_mean=want_mean, want_o32=True)
    else:
        o16, lse, mean = ops.attention(qkv, B, L, H, DH, want_mean=want_mean)
    x1 = torch.empty(M, E, device=dev, dtype=F32)
    ops.gemm(o16, pk.out_w, M, E, E, bias=pk.out_b, resid=x, out32=x1, round16=True)
    if tag:
        ops.KernelTimer.tag(None)
    _, a2 = ops.layernorm(x1, pk.ln2_w, pk.ln2_b, with_lo=ex)
    z = Split(torch.empty(M, 4 * E, device=dev, dtype=F16),
              torch.empty(M, 4 * E, device=dev, dtype=F16) if ex else None)
    u32 = torch.empty(M, 4 * E, device=dev, dtype=F32) if keep is not None else None
    ops.gemm(a2, pk.fc_w, M, 4 * E, E, bias=pk.fc_b, out16=z.hi, out16lo=z.lo, act=1, pre32=u32)
    x2 = torch.empty(M, E, device=dev, dtype=F32)
    x2h = None
    if x16_out is not None:     # fp16 copy of the block output for the adapter GEMMs (no extra pass)
        exo = ex or bool(int(config.head_lo) & 32)          # the adapters' input tokens carry their fp16 remainder
        nxt = x16_out.next_slot(M, E, dev, exo) if isinstance(x16_out, X16Stack) else None
        x2h = nxt or Split(torch.empty(M, E, device=dev, dtype=F16), torch.empty(M, E, device=dev, dtype=F16) if exo else None)
        x16_out.append(x2h)
    ops.gemm(z, pk.pj_w, M, E, 4 * E, bias=pk.pj_b, resid=x1, out32=x2,
             out16=x2h.hi if x2h else None, out16lo=x2h.lo if x2h else None)
    if keep is not None:
        keep.update(a32=a32, qkv=qkv, o32=o32, lse=lse, x1=x1, u32=u32)
    return x2, mean


def to_rows(x_lne):
    """(L, N, E) reference layout -> ((N*L, E) fp32 image-major rows, N, L)."""
    Lq, N, E = x_lne.shape
    return x_lne.detach().float().permute(1, 0, 2).contiguous().view(N * Lq, E), N, Lq


def from_rows(x, N, Lq):
    return x.view(N, Lq, -1).permute(1, 0, 2)
