"""Segmentation scores from a confusion histogram (reference utils/evaluate.py:10-36).

`scores` keeps the reference signature (lists of numpy label maps + a running histogram).  The
histogram itself can come from the device: `confusion_hist` counts (true, predicted) pairs of CUDA
label maps with the HIP kernel `wc_confusion_hist` (csrc/evalops.hip) into an int64 (nc, nc) tensor,
which the msc+flip driver all-reduces across ranks once at the end."""
import numpy as np
import torch


def _fast_hist(label_true, label_pred, num_classes):
    """hist[t, p] over the pixels whose true label is a class id (reference :10-16)."""
    lt = np.asarray(label_true).reshape(-1)
    lp = np.asarray(label_pred).reshape(-1)
    keep = (lt >= 0) & (lt < num_classes)
    idx = num_classes * lt[keep].astype(np.int64) + lp[keep].astype(np.int64)
    return np.bincount(idx, minlength=num_classes ** 2).reshape(num_classes, num_classes)


def confusion_hist(label_true, label_pred, num_classes, out=None):
    """Device version of `_fast_hist` summed over a batch: CUDA integer tensors of equal shape ->
    (nc, nc) int64 CUDA tensor (added into `out` if given).  Pixels whose true label is outside
    [0, nc) are skipped; a predicted label outside [0, nc) is an error in the reference too
    (np.bincount would grow the histogram): the kernel skips the pixel and raises a device flag that
    `check_predictions_in_range` reads back."""
    from .. import _lib as L
    L.require_gpu()
    lt = label_true.contiguous()
    lp = label_pred.contiguous()
    if lt.shape != lp.shape:
        raise RuntimeError("label_true and label_pred must have the same shape")
    if lt.dtype != torch.int64:
        lt = lt.long()
    if lp.dtype != torch.int64:
        lp = lp.long()
    if out is None:
        out = torch.zeros(num_classes, num_classes, device=lt.device, dtype=torch.int64)
    flag = _flags.get(lt.device)
    if flag is None:
        flag = _flags[lt.device] = torch.zeros(1, device=lt.device, dtype=torch.int32)
    L.lib().wc_confusion_hist(L.ptr(lt, torch.int64, "label_true"), L.ptr(lp, torch.int64, "label_pred"),
                              L.ptr(out, torch.int64, "hist"), L.ptr(flag, torch.int32, "flag"), lt.numel(),
                              int(num_classes), L.stream())
    return out


_flags = {}


def check_predictions_in_range(device):
    """True unless a confusion_hist call on `device` saw a predicted label outside [0, nc) since start-up (one
    device->host read; call it once after the evaluation loop, not per image)."""
    f = _flags.get(torch.device(device))
    return f is None or int(f.item()) == 0


def scores_from_hist(hist):
    """pAcc / mAcc / mIoU / per-class IoU of a confusion histogram (reference :23-36)."""
    hist = np.asarray(hist, dtype=np.float64)
    diag = np.diag(hist)
    with np.errstate(divide="ignore", invalid="ignore"):
        acc = diag.sum() / hist.sum()
        acc_cls = np.nanmean(diag / hist.sum(axis=1))
        iu = diag / (hist.sum(axis=1) + hist.sum(axis=0) - diag)
        valid = hist.sum(axis=1) > 0
        mean_iu = np.nanmean(iu[valid])
    return {"pAcc": acc, "mAcc": acc_cls, "miou": mean_iu, "iou": dict(zip(range(hist.shape[0]), iu))}


def scores(label_trues, label_preds, hist, num_classes=21):
    """Adds the pairs to `hist` in place and returns (hist, score dict), as the reference does."""
    for lt, lp in zip(label_trues, label_preds):
        hist += _fast_hist(lt, lp, num_classes)
    return hist, scores_from_hist(hist)


def pseudo_scores(label_trues, label_preds, num_classes=21):
    """Scores of pseudo labels where 255 in the prediction means "ignore" (reference :38-62)."""
    hist = np.zeros((num_classes, num_classes))
    for lt, lp in zip(label_trues, label_preds):
        lt = np.array(lt).reshape(-1)
        lp = np.array(lp).reshape(-1)
        lt[lp == 255] = 255
        lp[lp == 255] = 0
        hist += _fast_hist(lt, lp, num_classes)
    return scores_from_hist(hist)
