"""The training step around `WeCLIP.forward` (what reference scripts/dist_clip_voc.py:238-267 does
per iteration) and its data-parallel form: one process per GPU, image batch sharded over ranks,
one RCCL all-reduce of the adapter+decoder gradients (5,985,045 fp32 = 23.9 MB in ONE flat bucket;
the frozen CLIP encoder is replicated and never reduced).  The reference itself is single-GPU
(SURVEY.md §0-3); this is the DP launcher the build adds beside it.
"""
import os

import torch
import torch.distributed as dist
import torch.nn.functional as F

from .utils.camutils import cams_to_affinity_label, get_mask_by_radius
from .utils.losses import get_aff_loss, get_aff_loss_fused, get_seg_loss, get_seg_loss_fused
from .utils.optimizer import PolyWarmupAdamW


def make_optimizer(model, lr=2e-4, weight_decay=0.01, betas=(0.9, 0.999), warmup_iter=50, max_iter=30000,
                   warmup_ratio=1e-6, power=1.0):
    """configs/voc_attn_reg.yaml:29-38 + scripts/dist_clip_voc.py:196-234: only group 3 is non-empty
    and runs at 10x the base lr."""
    g = model.get_param_groups()
    groups = [
        {"params": g[0], "lr": lr, "weight_decay": weight_decay},
        {"params": g[1], "lr": 0.0, "weight_decay": 0.0},
        {"params": g[2], "lr": lr * 10, "weight_decay": weight_decay},
        {"params": g[3], "lr": lr * 10, "weight_decay": weight_decay},
    ]
    return PolyWarmupAdamW(groups, lr=lr, weight_decay=weight_decay, betas=list(betas),
                           warmup_iter=warmup_iter, max_iter=max_iter, warmup_ratio=warmup_ratio, power=power)


class GradBucket:
    """Flat fp32 bucket aliasing every trainable gradient, so the DP exchange is one collective."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        # every gradient starts on a 16-byte boundary (the fused AdamW and the split-K reductions use 16-byte accesses):
        # an odd-sized tensor (the 21-class bias) is followed by <= 3 padding elements, which stay zero
        offs, off = [], 0
        for p in self.params:
            offs.append(off)
            off = (off + p.numel() + 3) & ~3
        self.flat = torch.zeros(off, device=self.params[0].device, dtype=torch.float32)
        self.offsets = offs
        for p, o in zip(self.params, offs):
            p.grad = self.flat[o:o + p.numel()].view_as(p)

    def zero(self):
        self.flat.zero_()

    def all_reduce_mean(self, group=None):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            self.flat.div_(dist.get_world_size(group))


class TrainStep:
    """forward -> losses -> backward -> (DP: gradient all-reduce) -> AdamW step.

    graph=True (CUDA tensors + explicit `labels` only): forward + losses + backward of a batch signature
    (image shape, number of (image, class) pairs, max classes per image, affinity branch) are captured once into a
    HIP graph and replayed from then on -- one graph launch instead of ~420 kernel launches through Python, which is
    what the step costs on the host otherwise (BENCH_r01: 9.8 ms of enqueue per 14.5 ms step).  Inputs are copied
    into static buffers (images; the pair index tensors via PairPlan.update); the gradient all-reduce and the
    optimizer stay outside the graph (RCCL and the host-computed learning-rate schedule).  The first step of a new
    signature runs eagerly (it also warms every lazy initialisation), the second captures.

    pad_plans (default on in graph mode): the signature is BUCKETED (clip_tool.PairPlan pad=True: pair count to a multiple
    of 8 with dropped dummy pairs, channel capacity K to the next even number), so batches whose images carry 1..5
    classes share a few graphs instead of one per distinct (pair count, max classes); losses, gradients and parameters
    stay bit-identical to the unpadded eager step (tests/test_graph_step_gpu.py)."""

    def __init__(self, model, optimizer=None, radius=8, ignore_index=255, bucket=True, graph=False, pad_plans=True):
        self.model = model
        self.opt = optimizer or make_optimizer(model)
        self.radius, self.ignore = radius, ignore_index
        self._mask = {}
        self.bucket = GradBucket(model.get_param_groups()[3]) if bucket else None
        self.graph = bool(graph)
        self.pad_plans = bool(pad_plans)
        self._graphs = {}
        self._pool = None
        eng = getattr(model, "head_engine", None)
        if self.bucket is not None and eng is not None and os.environ.get("WECLIP_DIRECT_GRADS", "1") != "0":
            # the HIP head writes every adapter/decoder gradient straight into the bucket views
            # (the bucket is zeroed each step and each gradient is written exactly once per backward)
            eng.direct_grads = {n: p.grad for n, p in zip(eng.param_names(), eng.params()) if p.grad is not None}
        comer = getattr(model, "comer", None)
        if comer is not None:
            # the insert engine writes its parameter gradients into the views of THIS bucket (address range recorded: a .grad
            # that lies elsewhere is accumulated through autograd as usual); a TrainStep without a bucket clears the opt-in
            comer.direct_grads = False
            if self.bucket is not None and os.environ.get("WECLIP_DIRECT_GRADS", "1") != "0":
                lo = self.bucket.flat.data_ptr()
                comer.direct_grads = (lo, lo + 4 * self.bucket.flat.numel())
        if eng is not None and self.bucket is None:
            eng.direct_grads = None

    def mask(self, h, w, device):
        key = (h, w, str(device))
        if key not in self._mask:
            self._mask[key] = get_mask_by_radius(h, w, self.radius, device=device)
        return self._mask[key]

    def losses(self, seg, cam, attn_pred):
        h, w = cam.shape[1] // 16, cam.shape[2] // 16
        if seg.is_cuda:     # label->affinity-label + affinity loss, and up-sampling + both CE terms, fused (csrc/losses.hip)
            attn_loss = get_aff_loss_fused(attn_pred, cam, radius=self.radius, ignore_index=self.ignore)
            seg_loss = get_seg_loss_fused(seg, cam, ignore_index=self.ignore)
        else:
            aff_label = cams_to_affinity_label(cam, mask=self.mask(h, w, cam.device), ignore_index=self.ignore)
            attn_loss, _, _ = get_aff_loss(attn_pred, aff_label)
            segs = F.interpolate(seg, size=cam.shape[1:], mode="bilinear", align_corners=False)
            seg_loss = get_seg_loss(segs, cam.long(), ignore_index=self.ignore)
        return seg_loss + 0.1 * attn_loss, seg_loss, attn_loss

    def _reduce_grads(self):
        """One exchange per step: mean of the trainable gradients over the data-parallel ranks."""
        if self.bucket is not None:
            self.bucket.all_reduce_mean()
            return
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            # no flat bucket: reduce the gradients tensor by tensor (slower, same result) rather than let ranks diverge
            world = dist.get_world_size()
            for p in self.model.get_param_groups()[3]:
                if p.grad is not None:
                    dist.all_reduce(p.grad, op=dist.ReduceOp.SUM)
                    p.grad.div_(world)

    def _fwd_bwd(self, img, names, labels, plan=None):
        kw = {"plan": plan} if plan is not None else {}
        seg, cam, attn_pred = self.model(img, names if names is not None else [""] * img.shape[0], labels=labels, **kw)
        loss, seg_loss, attn_loss = self.losses(seg, cam, attn_pred)
        if self.bucket is not None:
            self.bucket.zero()
        else:
            self.opt.zero_grad()
        loss.backward()
        if img.is_cuda:      # backward nodes of a head that ran beside the CAM chain execute on the model's side stream
            cur = torch.cuda.current_stream()
            for s in getattr(self.model, "side_streams", lambda: [])():
                cur.wait_stream(s)
        return loss.detach(), seg_loss.detach(), attn_loss.detach()

    def __call__(self, img, names=None, labels=None):
        if self.graph and labels is not None and img.is_cuda and self.bucket is not None:
            out = self._graphed(img, labels)
        else:
            out = self._fwd_bwd(img, names, labels)
        self._reduce_grads()
        self.opt.step()
        return out

    # ---------------------------------------------------------------------------------- HIP-graph replay
    def _signature(self, img, labels):
        from .clip.clip_tool import PairPlan
        m = self.model
        seg_trans = (m.iter_num + 1) > m.seg_trans_after
        return (tuple(img.shape), PairPlan.signature(labels, self.pad_plans), bool(seg_trans), bool(m.training))

    def _graphed(self, img, labels):
        from .clip.clip_tool import PairPlan
        m = self.model
        sig = self._signature(img, labels)
        ent = self._graphs.get(sig)
        if ent is None:                      # first step of this signature: eager (warms lazy state), sets up the statics
            ent = self._graphs[sig] = {"graph": None, "img": torch.empty_like(img, dtype=torch.float32).contiguous(),
                                       "plan": PairPlan(labels, m.fg_text_features.shape[0], m.bg_text_features.shape[0],
                                                        img.device, pad=self.pad_plans)}
            ent["img"].copy_(img)
            return self._fwd_bwd(ent["img"], None, labels, plan=ent["plan"])
        ent["img"].copy_(img)
        ent["plan"].update(labels)
        if ent["graph"] is None:
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            it = m.iter_num
            # thread_local: other threads (RCCL's watchdog polling its events, the autograd workers allocating) must not
            # abort the capture; the kernels the autograd workers launch on the capturing stream are captured all the same
            with torch.cuda.graph(g, pool=self._pool, capture_error_mode="thread_local"):
                out = self._fwd_bwd(ent["img"], None, labels, plan=ent["plan"])
                ent["out"] = torch.stack(out)
            m.iter_num = it                  # the capture ran forward()'s counter once without executing a step
            self._pool = self._pool or g.pool()
            ent["graph"] = g
        ent["graph"].replay()
        m.iter_num += 1
        o = ent["out"].clone()               # the graph's pool memory is rewritten by the next replay
        return o[0], o[1], o[2]
