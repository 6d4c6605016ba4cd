"""Adapter head: same module tree / state-dict keys as reference WeCLIP_model/segformer_head.py
(`linears_modulelist.{i}.proj{,_2}`, `linear_fuse`, Dropout2d(0.1)); 11 x MLP(768->256, ReLU,
256->256) on the frozen feature maps, channel concat, 1x1 conv fuse.

Trainable.  On the GPU the training step does not go through these modules' `forward`: head_engine.HeadEngine runs the
eleven adapters + the fuse conv (and the decoder) forward AND backward as explicit HIP launches on the encoder's fp16 token
rows (grouped MFMA GEMMs, split-K weight gradients written straight into the all-reduce bucket); the modules own the
parameters and the state-dict keys.  The torch forms below are the CPU path and the `WECLIP_HEAD=torch` A/B path; the frozen
encoder features arrive as image-major token rows, so the (11,B,768,h,w) stack/permute copies of the reference
(model_attn_aff_voc.py:115-125) are skipped on the internal path (`forward_rows`).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class MLP(nn.Module):
    def __init__(self, input_dim=2048, embed_dim=768):
        super().__init__()
        self.proj = nn.Linear(input_dim, embed_dim)
        self.proj_2 = nn.Linear(embed_dim, embed_dim)

    def forward(self, x):
        """(n, c, h, w) -> (n, h*w, embed)"""
        return self.tokens(x.flatten(2).transpose(1, 2))

    def tokens(self, t):
        if t.is_cuda:          # MFMA GEMM forward / backward (hip_functional.py): used where the grouped HeadEngine is not
            from ..hip_functional import module_linear
            return module_linear(self.proj_2, module_linear(self.proj, t, act=2))
        return self.proj_2(F.relu(self.proj(t)))


class SegFormerHead(nn.Module):
    def __init__(self, in_channels=128, embedding_dim=256, num_classes=20, index=11, **kwargs):
        super().__init__()
        self.in_channels, self.num_classes, self.indexes = in_channels, num_classes, index
        c1 = in_channels[0] if isinstance(in_channels, (list, tuple)) else in_channels
        self.linears_modulelist = nn.ModuleList([MLP(c1, embedding_dim) for _ in range(index)])
        self.linear_fuse = nn.Conv2d(embedding_dim * index, embedding_dim, kernel_size=1)
        self.dropout = nn.Dropout2d(0.1)

    def _fuse(self, toks, n, h, w):
        cat = torch.cat(toks, dim=2)                                     # (n, hw, 256*index)
        y = F.linear(cat, self.linear_fuse.weight.flatten(1), self.linear_fuse.bias)
        y = y.transpose(1, 2).reshape(n, -1, h, w)
        return self.dropout(y)

    def forward(self, x_all):
        """x_all (index, n, c, h, w) -> (n, embedding_dim, h, w)   (reference signature)."""
        n, _, h, w = x_all.shape[1:]
        toks = [self.linears_modulelist[i](x_all[i].float()) for i in range(x_all.shape[0])]
        return self._fuse(toks, n, h, w)

    def forward_rows(self, rows_list, n, L, h, w):
        """rows_list: `index` tensors (n*L, c) fp32 image-major token rows (CLS first)."""
        toks = [self.linears_modulelist[i].tokens(r.view(n, L, -1)[:, 1:, :])
                for i, r in enumerate(rows_list)]
        return self._fuse(toks, n, h, w)
