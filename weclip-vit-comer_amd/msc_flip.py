"""Multi-scale + flip inference driver (BASELINE configs[4]): the per-image arithmetic of the
reference's `validate` (test_msc_flip_coco.py:33-121; the VOC twin test_msc_flip_voc.py) on the device.

Per image: resize the long side to `resize_long` (:52-57); run the model on [img, flip(img)] at scale 1 and on the
rescaled pair for every other scale (:60-86, `mode='val'`: the COCO model returns right after the decoder,
model_attn_aff_coco.py:131-132); un-flip and average each pair, bring the other scales to the scale-1 logit grid,
average over scales (:88); bilinear to the label size and arg-max, single-scale and multi-scale (:90-94); add
both predictions to (nc, nc) confusion histograms (utils/evaluate.py:10-16).  Nothing but the two int64 histograms
ever leaves the GPU; under data parallelism every rank evaluates its shard of the images (replicas, no collective
on the data path) and ONE all-reduce sums the histograms at the end.
"""
import torch
import torch.distributed as dist

from . import _lib as L
from .utils import evaluate

F32 = torch.float32


def scale_flip_pair(img, out_hw, scale_y, scale_x):
    """img (C,Hs,Ws) f32 -> (2,C,Hd,Wd): [bilinear resize, its horizontal flip].  scale_* = source step per
    destination pixel (F.interpolate(size=): in/out; F.interpolate(scale_factor=s): 1/s)."""
    C, Hs, Ws = img.shape
    Hd, Wd = out_hw
    out = torch.empty(2, C, Hd, Wd, device=img.device, dtype=F32)
    L.lib().wc_scale_flip_pair(L.ptr(img, F32, "img"), L.ptr(out), C, Hs, Ws, Hd, Wd, float(scale_y), float(scale_x), L.stream())
    return out


def flip_avg(segs, out, weight, accumulate):
    """out (C,Hd,Wd) (+)= weight * (R(segs[0]) + flip(R(segs[1]))) / 2, R = bilinear to out's grid."""
    _, C, Hs, Ws = segs.shape
    Hd, Wd = out.shape[1:]
    L.lib().wc_flip_avg(L.ptr(segs, F32, "segs"), L.ptr(out, F32, "out"), C, Hs, Ws, Hd, Wd, float(weight), 1 if accumulate else 0,
                        L.stream())
    return out


def resize_argmax(seg, out_hw):
    """argmax_c F.interpolate(seg[None], size=out_hw, bilinear)[0, c] -> (H, W) int64."""
    C, Hs, Ws = seg.shape
    pred = torch.empty(out_hw, device=seg.device, dtype=torch.int64)
    L.lib().wc_resize_argmax(L.ptr(seg, F32, "seg"), L.ptr(pred), C, Hs, Ws, out_hw[0], out_hw[1], L.stream())
    return pred


class MscFlipEvaluator:
    """`validate` of the reference, image by image.  model: WeCLIP (COCO or VOC) in eval mode on the GPU."""

    def __init__(self, model, num_classes, scales=(1.0, 0.75), resize_long=512):
        L.require_gpu()
        self.model, self.nc, self.resize_long = model, int(num_classes), resize_long
        self.scales = [float(s) for s in scales]
        dev = next(model.parameters()).device
        self.hist = torch.zeros(self.nc, self.nc, device=dev, dtype=torch.int64)        # single scale, no flip (`_preds`)
        self.msc_hist = torch.zeros(self.nc, self.nc, device=dev, dtype=torch.int64)    # multi-scale + flip (`_msc_preds`)
        self.images = 0

    @torch.no_grad()
    def logits(self, inputs):
        """inputs (1,3,H,W) -> (seg (nc,h,w): scale-1 un-flipped logits, msc (nc,h,w): multi-scale + flip average)."""
        x = inputs[0].float().contiguous()
        _, H, W = x.shape
        if self.resize_long:                                  # F.interpolate(size=(_h, _w)): scale = in / out
            ratio = self.resize_long / max(H, W)
            h1, w1 = int(H * ratio), int(W * ratio)
            pair = scale_flip_pair(x, (h1, w1), H / h1, W / w1)
        else:
            h1, w1 = H, W
            pair = scale_flip_pair(x, (H, W), 1.0, 1.0)
        base = pair[0]                                        # the resized, un-flipped input all other scales start from
        segs, _, _ = self.model(pair, ["", ""], mode="val")
        segs = segs.float().contiguous()
        seg1 = segs[0].contiguous()
        msc = torch.empty_like(seg1)
        w = 1.0 / (1 + sum(1 for s in self.scales if s != 1.0))
        flip_avg(segs, msc, w, accumulate=False)
        for s in self.scales:
            if s == 1.0:
                continue
            hs, ws = int(h1 * s), int(w1 * s)                 # F.interpolate(scale_factor=s): floor(size * s), step 1/s
            pair_s = scale_flip_pair(base, (hs, ws), 1.0 / s, 1.0 / s)
            segs_s, _, _ = self.model(pair_s, ["", ""], mode="val")
            flip_avg(segs_s.float().contiguous(), msc, w, accumulate=True)
        return seg1, msc

    @torch.no_grad()
    def add(self, inputs, labels):
        """One image: inputs (1,3,H,W) normalised pixels, labels (1,Hl,Wl) integer class map (255 = ignore).
        Returns (seg_pred, msc_pred) (Hl,Wl) int64 on the device and updates both histograms."""
        seg1, msc = self.logits(inputs.cuda())
        lab = labels[0].cuda().long().contiguous()
        seg_pred = resize_argmax(seg1, tuple(lab.shape))
        msc_pred = resize_argmax(msc, tuple(lab.shape))
        evaluate.confusion_hist(lab, seg_pred, self.nc, out=self.hist)
        evaluate.confusion_hist(lab, msc_pred, self.nc, out=self.msc_hist)
        self.images += 1
        return seg_pred, msc_pred

    def reduce(self, group=None):
        """Sum the histograms over the data-parallel ranks (one small int64 all-reduce each)."""
        reduce_hist(self.hist, group)
        reduce_hist(self.msc_hist, group)

    def scores(self):
        return evaluate.scores_from_hist(self.hist.cpu().numpy()), evaluate.scores_from_hist(self.msc_hist.cpu().numpy())


def reduce_hist(hist, group=None):
    """In-place SUM of an int64 histogram over the ranks (RCCL on GPUs, gloo in the CPU tests)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(hist, op=dist.ReduceOp.SUM, group=group)
    return hist


def shard(items, rank, world):
    """Rank r evaluates items r, r + world, ... (replicas only: images are independent units)."""
    return list(items)[rank::world]
