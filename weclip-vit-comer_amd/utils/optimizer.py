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
        out = super().step(closure)
        self.global_step += 1
        return out
