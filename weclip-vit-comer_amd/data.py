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
    from its own seed stream, so no two ranks (and no two consecutive steps) see the same tensor.

    source="float": the pool holds ready normalised batches (what the parity tests feed).
    source="uint8": the pool holds uint8 (B, 375, 500, 3) "decoded JPEGs" on the device and every next() runs the
    device-side input pipeline (DeviceAugment: random rescale / flip / pad + crop / normalise, csrc/augment.hip) --
    the per-step work of the reference's loader, minus JPEG decoding, on the GPU and inside the timed step."""

    def __init__(self, batch, size, classes_per_image=2, rank=0, world=1, seed=100, pool=4, device="cuda",
                 n_classes=20, source="float", src_hw=(375, 500)):
        self.batch, self.size, self.rank, self.world, self.source = batch, size, rank, world, source
        self.images, self.labels = [], []
        for j in range(pool):
            s = seed + 1000 * j + rank            # j = 0, rank = 0 is bench.py's historical batch (seed 100 / 7)
            if source == "uint8":
                f = synth.make_images(batch, src_hw[0], src_hw[1], seed=s)
                u8 = (f * 58.0 + 118.0).clamp_(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
                self.images.append(u8.to(device))
            else:
                self.images.append(synth.make_images(batch, size, size, seed=s).to(device))
            self.labels.append(synth.make_label_lists(batch, classes_per_image, n_classes=n_classes,
                                                      seed=7 + 1000 * j + rank))
        self.aug = DeviceAugment(crop_size=size, seed=seed + rank) if source == "uint8" else None
        self._i = 0

    def __len__(self):
        return len(self.images)

    def next(self):
        i = self._i % len(self.images)
        self._i += 1
        if self.aug is not None:
            return self.aug(self.images[i]), self.labels[i]
        return self.images[i], self.labels[i]

    def __iter__(self):
        while True:
            yield self.next()


MEAN = (123.675, 116.28, 103.53)       # datasets/transforms.py:8
STD = (58.395, 57.12, 57.375)


class DeviceAugment:
    """The reference's train-time augmentation (datasets/voc.py:108-143: random_scaling -> random_fliplr -> random_crop
    -> normalize_img -> CHW) with the random draws on the host and the pixel work in HIP kernels (csrc/augment.hip):
    uint8 (B,H,W,3) on the device in, float32 (B,3,crop,crop) normalised out, the rescale being Pillow's BILINEAR for
    8-bit images bit for bit (up- and down-scaling).  The host never touches a pixel, so the loader cannot serialise
    the step.

    The draws follow the reference's own sources and order per image -- `random.uniform` (scale, transforms.py:31),
    `random.random` (flip, :71), `np.random.randint` x 2 (pad offsets, :134-135), `random.randrange` x 2 (crop box,
    :143-146; without a label map the first box is taken) -- from private `random.Random(seed)` /
    `np.random.RandomState(seed)` instances, so a loader seeded like the reference's worker produces its crops."""

    def __init__(self, crop_size=512, rescale_range=(0.5, 2.0), fliplr=True, seed=0, mean=MEAN, std=STD, np_seed=None):
        import random
        import numpy as np
        self.crop, self.range, self.fliplr = int(crop_size), tuple(rescale_range) if rescale_range else None, bool(fliplr)
        self.py_rng = random.Random(seed)
        self.np_rng = np.random.RandomState(seed if np_seed is None else np_seed)
        self.mean, self.std = tuple(float(v) for v in mean), tuple(float(v) for v in std)
        self._ws = None

    def draw_one(self, H, W):
        """(scale, flip, rh, rw, pad_y, pad_x, crop_y, crop_x) of one image."""
        s = self.py_rng.uniform(*self.range) if self.range else 1.0
        rh, rw = (int(s * H), int(s * W)) if self.range else (H, W)
        flip = int(self.py_rng.random() > 0.5) if self.fliplr else 0
        ch, cw = max(self.crop, rh), max(self.crop, rw)                  # canvas (transforms.py:123-124)
        pad_y, pad_x = int(self.np_rng.randint(ch - rh + 1)), int(self.np_rng.randint(cw - rw + 1))
        crop_y = self.py_rng.randrange(0, ch - self.crop + 1, 1)
        crop_x = self.py_rng.randrange(0, cw - self.crop + 1, 1)
        return s, flip, rh, rw, pad_y, pad_x, crop_y, crop_x

    @staticmethod
    def pack(draws):
        """List of draw_one() tuples -> int32 tensor (B, 8) in the kernel's record layout."""
        import numpy as np
        rec = np.zeros((len(draws), 8), np.int32)
        for b, d in enumerate(draws):
            rec[b, 0] = np.float32(d[0]).view(np.int32)
            rec[b, 1:] = d[1:]
        return torch.from_numpy(rec)

    def draw(self, B, H, W):
        """Host-side random parameters of one batch -> int32 tensor (B, 8) in the kernel's record layout."""
        return self.pack([self.draw_one(H, W) for _ in range(B)])

    def __call__(self, images_u8, params=None):
        """images_u8 (B,H,W,3) uint8 CUDA; params: a draw() result (default: a fresh draw).  -> (B,3,crop,crop) f32."""
        import ctypes
        from . import _lib as L
        L.require_gpu()
        B, H, W, C = images_u8.shape
        if C != 3 or images_u8.dtype != torch.uint8:
            raise RuntimeError("DeviceAugment expects uint8 (B, H, W, 3) images")
        if params is None:
            params = self.draw(B, H, W)
        if not params.is_cuda:
            rhw = params[:, 2:4]
            if int(rhw.min()) < 1 or H > 4 * int(rhw[:, 0].min()) or W > 4 * int(rhw[:, 1].min()):
                raise RuntimeError("DeviceAugment: rescaled size must be >= 1 and down-scaling at most 4x")
        p = params.pin_memory().to(images_u8.device, non_blocking=True) if not params.is_cuda else params
        out = torch.empty(B, 3, self.crop, self.crop, device=images_u8.device, dtype=torch.float32)
        n = ctypes.c_long(0)
        L.lib().wc_augment_workspace_ints(B, self.crop, ctypes.byref(n))
        if self._ws is None or self._ws.numel() < n.value or self._ws.device != images_u8.device:
            self._ws = torch.empty(n.value, device=images_u8.device, dtype=torch.int32)
        L.lib().wc_augment_normalize(L.ptr(images_u8.contiguous(), torch.uint8, "images"), L.ptr(p, torch.int32, "params"),
                                     L.ptr(out), L.ptr(self._ws, torch.int32), B, H, W, self.crop,
                                     (ctypes.c_float * 3)(*self.mean), (ctypes.c_float * 3)(*self.std), L.stream())
        return out
