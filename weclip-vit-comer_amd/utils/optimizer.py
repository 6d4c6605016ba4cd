"""`PolyWarmupAdamW` (reference utils/optimizer.py:3-33): AdamW whose per-group lr follows a
linear warm-up then a polynomial decay, written into `param_groups[i]['lr']` before each step."""
import torch


class PolyWarmupAdamW(torch.optim.AdamW):
    def __init__(self, params, lr, weight_decay, betas, warmup_iter=None, max_iter=None,
                 warmup_ratio=None, power=None):
        super().__init__(params, lr=lr, betas=betas, weight_decay=weight_decay, eps=1e-8)
        self.global_step = 0
        self.warmup_iter, self.warmup_ratio = warmup_iter, warmup_ratio
        self.max_iter, self.power = max_iter, power
        self._base_lr = [g["lr"] for g in self.param_groups]

    def _lr_mult(self):
        s = self.global_step
        if s < self.warmup_iter:
            return 1 - (1 - s / self.warmup_iter) * (1 - self.warmup_ratio)
        if s < self.max_iter:
            return (1 - s / self.max_iter) ** self.power
        return None

    def step(self, closure=None):
        m = self._lr_mult()
        if m is not None:
            for g, base in zip(self.param_groups, self._base_lr):
                g["lr"] = base * m
        if closure is None and self._hip_ok():
            out = self._hip_step()
        else:
            out = super().step(closure)
        self.global_step += 1
        return out

    # ---- one-launch-per-group HIP path (csrc/train_ops.hip adamw_multi_kernel) --------------------------------
    def _hip_ok(self):
        for g in self.param_groups:
            if g.get("amsgrad") or g.get("maximize") or g.get("capturable") or g.get("differentiable"):
                return False
            for p in g["params"]:
                if p.grad is not None and not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()
                                               and p.grad.is_cuda and p.grad.dtype == torch.float32
                                               and p.grad.is_contiguous() and not p.grad.is_sparse):
                    return False
        return True

    @torch.no_grad()
    def _hip_step(self):
        from .. import _lib as L
        if not hasattr(self, "_hip"):
            self._hip = {}
        for gi, group in enumerate(self.param_groups):
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            for p in params:
                s = self.state[p]
                if len(s) == 0:            # same state layout as torch.optim.AdamW (checkpoints interoperate)
                    s["step"] = torch.tensor(0.0, dtype=torch.float32)
                    s["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    s["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            # pointer table, rebuilt whenever a tensor was replaced (first step, load_state_dict, new .grad)
            key = tuple((p.data_ptr(), p.grad.data_ptr(), self.state[p]["exp_avg"].data_ptr(),
                         self.state[p]["exp_avg_sq"].data_ptr(), p.numel()) for p in params)
            st = self._hip.get(gi)
            if st is None or st["key"] != key:
                dev = params[0].device
                rows = []
                for p in params:
                    s = self.state[p]
                    rows.append([p.data_ptr(), p.grad.data_ptr(), s["exp_avg"].data_ptr(), s["exp_avg_sq"].data_ptr(),
                                 p.numel(), 0, 0, 0])
                st = self._hip[gi] = {"key": key, "count": len(rows),
                                      "table": torch.tensor(rows, dtype=torch.int64).pin_memory().to(dev, non_blocking=True)}
            for p in params:
                self.state[p]["step"] += 1
            step = float(self.state[params[0]]["step"])
            b1, b2 = group["betas"]
            L.lib().wc_adamw_multi(L.ptr(st["table"], torch.int64, "table"), st["count"], float(group["lr"]), float(b1),
                                   float(b2), float(group["eps"]), float(group["weight_decay"]), 1.0 - b1 ** step,
                                   1.0 - b2 ** step, 64, L.stream())
            inc = getattr(torch.autograd.graph, "increment_version", None)
            if inc is not None:            # the update happened below torch's view: bump the version counters
                for p in params:           # (autograd's saved-tensor checks, ops.WeightCache)
                    inc(p)
        return None
