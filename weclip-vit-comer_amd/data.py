"""Synthetic VOC-shaped input for the data-parallel step: per-rank seeded batches that
differ from step to step, resident on the device.

The reference's loader (datasets/voc.py:190-250 + datasets/transforms.py) decodes JPEGs on the host, rescales /
flips / crops with numpy + PIL and normalises with mean/std (transforms.py:8-15); its output per step is a float32
(B, 3, crop, crop) tensor of ~N(0,1) pixels plus the image-level class ids.  No dataset exists offline, so this
loader draws that output directly: a pool of `pool` distinct batches per rank (seed = f(base seed, rank)), cycled.
"""
import torch

from . import synth


class SyntheticVOCLoader:
    """Iterable of (images (B,3,S,S) f32 CUDA, label lists) -- `DistributedSampler`-like: rank r of `world` draws
    from its own seed stream, so no two ranks (and no two consecutive steps) see the same tensor."""

    def __init__(self, batch, size, classes_per_image=2, rank=0, world=1, seed=100, pool=4, device="cuda",
                 n_classes=20):
        self.batch, self.size, self.rank, self.world = batch, size, rank, world
        self.images, self.labels = [], []
        for j in range(pool):
            s = seed + 1000 * j + rank            # j = 0, rank = 0 is bench.py's historical batch (seed 100 / 7)
            self.images.append(synth.make_images(batch, size, size, seed=s).to(device))
            self.labels.append(synth.make_label_lists(batch, classes_per_image, n_classes=n_classes,
                                                      seed=7 + 1000 * j + rank))
        self._i = 0

    def __len__(self):
        return len(self.images)

    def next(self):
        i = self._i % len(self.images)
        self._i += 1
        return self.images[i], self.labels[i]

    def __iter__(self):
        while True:
            yield self.next()
