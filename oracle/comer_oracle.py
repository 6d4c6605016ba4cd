"""CPU restatement of multi-scale deformable attention (TEST INFRASTRUCTURE).

The reference repository contains NO ViT-CoMer code: this oracle restates
the published MSDeformAttn core that ViT-CoMer's CTI blocks are built on (ViT_CoMer.pdf §3.3; Deformable
DETR eq. 2-3) with torch's grid_sample.  PARITY UNPINNED: no reference function, test or golden vector
exists for it; the HIP kernels (csrc/msdeform.hip) are checked against this file only.
"""
import torch
import torch.nn.functional as F


def ms_deform_attn(value, spatial_shapes, sampling_locations, attention_weights):
    """value (N,S,M,D); spatial_shapes [(H,W)...]; sampling_locations (N,Lq,M,nL,nP,2) in [0,1] (x,y);
    attention_weights (N,Lq,M,nL,nP).  Returns (N,Lq,M*D)."""
    N, S, M, D = value.shape
    _, Lq, _, nL, nP, _ = sampling_locations.shape
    vals = value.split([h * w for h, w in spatial_shapes], dim=1)
    grids = 2 * sampling_locations - 1
    samples = []
    for l, (h, w) in enumerate(spatial_shapes):
        v = vals[l].flatten(2).transpose(1, 2).reshape(N * M, D, h, w)
        g = grids[:, :, :, l].transpose(1, 2).flatten(0, 1)                       # (N*M, Lq, nP, 2)
        samples.append(F.grid_sample(v, g, mode="bilinear", padding_mode="zeros", align_corners=False))
    aw = attention_weights.transpose(1, 2).reshape(N * M, 1, Lq, nL * nP)
    out = (torch.stack(samples, dim=-2).flatten(-2) * aw).sum(-1).view(N, M * D, Lq)
    return out.transpose(1, 2).contiguous()
