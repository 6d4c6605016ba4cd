"""Live loss functions of the reference training step (utils/losses.py:11-22 get_aff_loss;
scripts/dist_clip_voc.py:105-113 get_seg_loss).  Stock PyTorch-ROCm ops."""
import torch
import torch.nn.functional as F


def get_aff_loss(inputs, targets):
    pos = targets == 1
    neg = targets == 0
    pos_count = pos.sum() + 1
    neg_count = neg.sum() + 1
    pos_loss = (pos * (1 - inputs)).sum() / pos_count
    neg_loss = (neg * inputs).sum() / neg_count
    return 0.5 * pos_loss + 0.5 * neg_loss, pos_count, neg_count


def get_seg_loss(pred, label, ignore_index=255):
    bg = label.clone()
    bg[label != 0] = ignore_index
    fg = label.clone()
    fg[label == 0] = ignore_index
    return 0.5 * (F.cross_entropy(pred, bg.long(), ignore_index=ignore_index)
                  + F.cross_entropy(pred, fg.long(), ignore_index=ignore_index))
