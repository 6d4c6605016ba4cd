"""Running means keyed by name (reference utils/AverageMeter.py:1-30); host-side bookkeeping of the
training script, kept so `from utils.AverageMeter import AverageMeter` resolves in drop-in mode."""


class AverageMeter:
    def __init__(self, *keys):
        self._sum = {k: 0.0 for k in keys}
        self._cnt = {k: 0 for k in keys}

    def add(self, values):
        for k, v in values.items():
            self._sum[k] = self._sum.get(k, 0.0) + v
            self._cnt[k] = self._cnt.get(k, 0) + 1

    def get(self, *keys):
        means = tuple(self._sum[k] / self._cnt[k] for k in keys)      # ZeroDivisionError on an empty key, like the reference
        return means[0] if len(means) == 1 else means

    def pop(self, key=None):
        if key is None:
            for k in self._sum:
                self._sum[k], self._cnt[k] = 0.0, 0
            return None
        v = self.get(key)
        self._sum[key], self._cnt[key] = 0.0, 0
        return v
