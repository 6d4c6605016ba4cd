"""`cams_to_affinity_label` (reference utils/camutils.py:226-247) and `get_mask_by_radius`
(scripts/dist_clip_voc.py:116-133), vectorised; stock PyTorch ops."""
import torch
import torch.nn.functional as F


def get_mask_by_radius(h=20, w=20, radius=8, device="cpu"):
    """(hw, hw) 0/1 mask of token pairs within a Chebyshev window of `radius`."""
    idx = torch.arange(h * w, device=device)
    yy, xx = idx // w, idx % w
    m = ((yy[:, None] - yy[None]).abs() <= radius) & ((xx[:, None] - xx[None]).abs() <= radius)
    return m.to(torch.float32)


def cams_to_affinity_label(cam_label, mask=None, ignore_index=255):
    b, h, w = cam_label.shape
    r = F.interpolate(cam_label.unsqueeze(1).float(), size=[h // 16, w // 16], mode="nearest")
    lab = r.reshape(b, 1, -1)
    aff = (lab.transpose(1, 2) == lab).long()
    ign = lab == ignore_index
    bad = ign | ign.transpose(1, 2)
    if mask is not None:
        bad = bad | (torch.as_tensor(mask, device=cam_label.device)[None] == 0)
    aff[bad] = ignore_index
    return aff
