"""Per-image CAM driver API of reference clip/clip_tool.py, executed by the batched device
pipeline (gradcam_engine + cam_pipeline): nothing leaves the GPU between GradCAM, the affinity
refinement and the label map."""
import os

import numpy as np
import torch

from .. import cam_pipeline as CP
from ..gradcam_engine import last_layer_forward
from . import vit_engine as VE

I32 = torch.int32


class ClipOutputTarget:
    def __init__(self, category):
        self.category = category

    def __call__(self, model_output):
        return model_output[self.category] if model_output.dim() == 1 else model_output[:, self.category]


def generate_clip_fts(image, model, require_all_fts=True):
    if image.dim() == 3:
        image = image.unsqueeze(0)
    h, w = image.shape[-2:]
    return model.encode_image(image.cuda(), h, w, require_all_fts=require_all_fts)


def compute_trans_mat(attn_weight):
    """(hw, hw) affinity -> Sinkhorn x3, symmetrise, square (reference clip_tool.py:64-80)."""
    from .. import torch_ops  # noqa: F401  (registers torch.ops.weclip.*)
    return torch.ops.weclip.trans_mat(attn_weight.detach().float().contiguous()[None].cuda())[0]


def read_image_labels(img_path):
    """Image-level class ids from the GT mask PNG (reference clip_tool.py:111-124): unique pixel
    values minus one (uint8 arithmetic, so background 0 -> 255), 255/254 dropped.
    Returns (ids, (orig_h, orig_w))."""
    from PIL import Image
    arr = np.asarray(Image.open(img_path))
    ids = (np.unique(arr) - 1).tolist()
    ids = [int(i) for i in ids if i not in (255, 254)]
    return ids, arr.shape[:2]


class PairPlan:
    """Flattened (image, class) pairs of a batch and the index tensors the kernels consume.

    The device tensors are allocated once; `update(label_lists)` refills them in place for another batch with the
    same signature (B, P, K) -- what a captured HIP graph of the step needs (train_step.TrainStep(graph=True)):
    the graph's kernels keep reading the same addresses, only their contents change.

    pad=True buckets the signature so that real label distributions (the reference takes 1..5+ classes per image from
    the GT mask, clip_tool.py:111-124) need only a handful of captured graphs: the pair count is padded to a multiple
    of PAD_P with dummy pairs (copies of pair 0 whose slot is -1: their GradCAM is computed and dropped, box_mask skips
    the write) and K, the channel capacity per image, to the next even number; per-image class counts (nk / nch) stay
    exact, so every real pair and channel sees the arithmetic of the unpadded plan."""
    PAD_P = 8

    def __init__(self, label_lists, n_fg, n_bg, device, pad=False):
        self.pad = bool(pad)
        self.n_fg, self.n_bg, self.device = n_fg, n_bg, torch.device(device)
        host, vk = self._host_arrays(label_lists)
        cuda = self.device.type == "cuda"
        # ONE pinned, asynchronous host->device copy for all the int32 index arrays (a pageable
        # torch.tensor(..., device=cuda) per array is a synchronising copy: it stalls the launch queue)
        self._host = host.pin_memory() if cuda else host
        self._host_vk = vk.pin_memory() if cuda else vk
        self._dev_all = self._host.to(self.device, non_blocking=True)
        self.valid_key = self._host_vk.to(self.device, non_blocking=True)
        views, off = [], 0
        for n in self._part_len:
            views.append(self._dev_all[off:off + n])
            off += n
        self.pair_img, self.pair_cls, self.pair_slot, text_idx, self.n_text, self.nk, self.nch = views
        self.text_idx = text_idx.view(self.P, self.Tmax)

    @staticmethod
    def signature(label_lists, pad=False):
        """(B, P, K): batches with equal signatures can share one set of device index tensors."""
        P, K = sum(len(l) for l in label_lists), max(len(l) for l in label_lists)
        if pad:
            P, K = -(-P // PairPlan.PAD_P) * PairPlan.PAD_P, max(2, K + (K & 1))
        return len(label_lists), P, K

    def _host_arrays(self, label_lists):
        self.label_lists = [list(map(int, l)) for l in label_lists]
        if any(len(l) == 0 for l in self.label_lists):
            raise RuntimeError("every image needs at least one foreground class id")
        self.B, P_sig, self.K = self.signature(self.label_lists, self.pad)
        self.Tmax = self.K + self.n_bg
        pi, pc, ps, ti, nt = [], [], [], [], []
        for i, ids in enumerate(self.label_lists):
            rows = ids + [self.n_fg + r for r in range(self.n_bg)]
            for j in range(len(ids)):
                pi.append(i); pc.append(j); ps.append(j)
                ti.append(rows + [0] * (self.Tmax - len(rows)))
                nt.append(len(rows))
        self.P_real = len(pi)
        for _ in range(P_sig - len(pi)):          # dummy pairs (pad=True): pair 0 again, result dropped (slot -1)
            pi.append(pi[0]); pc.append(pc[0]); ps.append(-1)
            ti.append(list(ti[0]))
            nt.append(nt[0])
        nk = [len(l) for l in self.label_lists]
        flat_ti = [v for row in ti for v in row]
        parts = [pi, pc, ps, flat_ti, nt, nk, [k + 1 for k in nk]]
        self._part_len = [len(p) for p in parts]
        self.P = len(pi)
        host = torch.tensor([v for part in parts for v in part], dtype=I32)
        vk = torch.zeros(self.B, self.K + 1, dtype=torch.int64)
        for i, ids in enumerate(self.label_lists):
            vk[i, 1:1 + len(ids)] = torch.tensor(ids) + 1
        return host, vk

    def update(self, label_lists):
        """Refill the device index tensors for another batch of the same signature (asynchronous copies from the
        pinned staging buffers, ordered on the current stream)."""
        if self.signature(label_lists, self.pad) != (self.B, self.P, self.K):
            raise RuntimeError(f"PairPlan.update: signature {self.signature(label_lists, self.pad)} != {(self.B, self.P, self.K)}")
        host, vk = self._host_arrays(label_lists)
        if self.device.type != "cuda":
            self._dev_all.copy_(host)
            self.valid_key.copy_(vk)
            return self
        # two pinned staging sets used alternately, each guarded by the event recorded after its last copy: the host
        # only waits if it is two updates ahead of the device (never a full-stream synchronisation per step)
        if not hasattr(self, "_ring"):
            self._ring = [(self._host, self._host_vk, None), (self._host.clone().pin_memory(), self._host_vk.clone().pin_memory(), None)]
            self._slot = 0
        self._slot ^= 1
        h, v, ev = self._ring[self._slot]
        if ev is not None:
            ev.synchronize()
        h.copy_(host)
        v.copy_(vk)
        self._dev_all.copy_(h, non_blocking=True)
        self.valid_key.copy_(v, non_blocking=True)
        ev = ev or torch.cuda.Event()
        ev.record()
        self._ring[self._slot] = (h, v, ev)
        return self


def normalised_text(fg_text, bg_text, device):
    t = torch.cat([fg_text.detach().float(), bg_text.detach().float()], 0).to(device)
    return (t / t.norm(dim=1, keepdim=True)).contiguous()


def batch_refined_cams(clip_model, last_rows, maps11, seg_attn, plan, text_hat, h, w, thr,
                       seg_trans, n_last, keep=None):
    """GradCAM for every pair + affinity refinement.  last_rows (B*L, E): block layers-1 output;
    maps11: list of that many head-mean maps (entries may be None where unused).
    Returns (R (B, hw, K) refined CAMs, cams (P, hw), probs (P, Tmax), state)."""
    B, Lq = plan.B, h * w + 1
    st = last_layer_forward(clip_model, last_rows, B, Lq)
    maps = list(maps11) + [st.mean]
    need = maps[-n_last:] if seg_trans else maps[-8:]
    if any(m is None for m in need):
        raise RuntimeError("attention maps needed by the affinity were not computed")
    if callable(seg_attn):      # attn_pred still being computed on another stream (WeCLIP.forward): join it only now
        if seg_trans:
            cams, probs, _ = st.grad_cam(text_hat, plan.text_idx, plan.n_text, plan.pair_img, plan.pair_cls, plan.Tmax)
            W, c1 = CP.affinity_weight(maps, seg_attn(), seg_trans, n_last, keep=keep, return_c1=True)
            R = CP.refine(W, cams, plan.pair_img, plan.pair_slot, plan.K, h, w, thr, c1=c1)
            return R, cams, probs, st
        seg_attn = None
    # (the affinity weight beside the GradCAM GEMMs on a second stream was measured neutral to slower in round 3: one stream)
    cams, probs, _ = st.grad_cam(text_hat, plan.text_idx, plan.n_text, plan.pair_img, plan.pair_cls, plan.Tmax)
    W, c1 = CP.affinity_weight(maps, seg_attn, seg_trans, n_last, keep=keep, return_c1=True)
    R = CP.refine(W, cams, plan.pair_img, plan.pair_slot, plan.K, h, w, thr, c1=c1)
    return R, cams, probs, st


_SIDE = {}


def side_stream(dev):
    """The package's second HIP stream on `dev` (head forward beside the CAM chain)."""
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=dev)
    return _SIDE[key]


def side_streams():
    return list(_SIDE.values())


def _single(img_path, image, image_features, attn_weight_list, seg_attn, bg_text_features, fg_text_features,
            cam, mode, require_seg_trans, thr, n_last, labels=None):
    if labels is None:
        ids, (oh, ow) = read_image_labels(img_path)
    else:
        ids, (oh, ow) = list(labels), image.shape[-2:]
    H, W = image.shape[-2:]
    h, w = H // 16, W // 16
    dev = image_features.device
    rows, N, Lq = VE.to_rows(image_features)
    plan = PairPlan([ids], fg_text_features.shape[0], bg_text_features.shape[0], dev)
    text_hat = normalised_text(fg_text_features, bg_text_features, dev)
    maps = [m[None].float().contiguous() for m in attn_weight_list]
    seg = seg_attn.reshape(1, h * w, h * w) if seg_attn is not None else None
    R, _, _, _ = batch_refined_cams(cam.model, rows, maps, seg, plan, text_hat, h, w, thr, require_seg_trans, n_last)
    refined = [R[0, :, k].reshape(h, w) for k in range(len(ids))]
    return (refined, ids, W, H) if mode == "train" else (refined, ids, ow, oh)


def perform_single_voc_cam(img_path, image, image_features, attn_weight_list, seg_attn, bg_text_features,
                           fg_text_features, cam, mode="train", require_seg_trans=False, labels=None):
    """Reference clip_tool.py:106-197.  `cam` is this package's GradCAM (its .model is used)."""
    return _single(img_path, image, image_features, attn_weight_list, seg_attn, bg_text_features,
                   fg_text_features, cam, mode, require_seg_trans, 0.4, 6, labels)


def perform_single_coco_cam(img_path, image, image_features, attn_weight_list, seg_attn, bg_text_features,
                            fg_text_features, cam, mode="train", require_all_fts=True, require_seg_trans=False,
                            labels=None):
    """Reference clip_tool.py:221-319 (threshold 0.7, last 10 maps in the seg-trans branch)."""
    return _single(img_path, image, image_features, attn_weight_list, seg_attn, bg_text_features,
                   fg_text_features, cam, mode, require_seg_trans, 0.7, 10, labels)


def generate_cam_label(cam_refined_list, keys, w, h):
    """Reference clip_tool.py:202-216: min-max + bilinear to (h, w) of every refined CAM."""
    R = torch.stack([c.reshape(-1) for c in cam_refined_list], dim=1)[None].float().contiguous().cuda()
    hh, ww = cam_refined_list[0].shape
    nk = torch.tensor([len(cam_refined_list)], dtype=I32, device="cuda")
    cams = CP.upsample_with_bg(R, nk, hh, ww, h, w)
    return {"keys": np.asarray(keys), "refined_cam": cams[0, 1:]}
