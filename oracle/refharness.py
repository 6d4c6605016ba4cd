"""Import the UNMODIFIED reference (/root/reference) on CPU -- build-container only.

TEST INFRASTRUCTURE.  Used only by tests/golden/make_golden.py (fixture generation) and by
the optional live cross-checks in tests/ that skip when /root/reference is absent.  Nothing
here is copied from the reference; it only arranges for the reference's own modules to be
importable without the third-party packages this image lacks (SURVEY.md §8c work-arounds):

  * torchvision / ftfy / lxml / ttach / mmcv / tqdm: empty or identity stubs (import-only use);
  * cv2: a *functional* stand-in for the four calls on the path (resize, threshold,
    findContours, boundingRect) built from torch / scipy per the OpenCV documentation.
    It could not be compared with real OpenCV here => the box step is "unverified vs cv2";
  * Tensor.cuda / Module.cuda rebound to identity so `.cuda()` call sites run on CPU.

The GPU box never sees /root/reference, so nothing under `-m gpu`, smoke() or bench.py
imports this file.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"


def available():
    return os.path.isdir(os.path.join(REF, "clip"))


def _cv2_stub():
    import scipy.ndimage as ndi
    import torch.nn.functional as F
    cv2 = types.ModuleType("cv2")
    cv2.__version__ = "4.6.0-stub"
    cv2.THRESH_BINARY = 0
    cv2.RETR_TREE = 3
    cv2.CHAIN_APPROX_SIMPLE = 2
    cv2.INTER_LINEAR = 1
    cv2.COLORMAP_JET = 2
    cv2.COLOR_BGR2RGB = 4

    def resize(src, dsize, *a, **k):
        w, h = dsize
        t = torch.from_numpy(np.ascontiguousarray(src, dtype=np.float32))[None, None]
        return F.interpolate(t, size=(h, w), mode="bilinear", align_corners=False)[0, 0].numpy()

    def threshold(src, thresh, maxval, type):
        return thresh, ((src > thresh) * maxval).astype(np.uint8)

    def findContours(image, mode, method):
        img = np.asarray(image)
        if img.ndim == 3:
            img = img[..., 0]
        lab, n = ndi.label(img > 0, structure=np.ones((3, 3)))
        cs = []
        for i in range(1, n + 1):
            ys, xs = np.nonzero(lab == i)
            cs.append(np.stack([xs, ys], axis=1)[:, None, :].astype(np.int32))
        return cs, None

    def boundingRect(c):
        xs, ys = c[:, 0, 0], c[:, 0, 1]
        return int(xs.min()), int(ys.min()), int(xs.max() - xs.min() + 1), int(ys.max() - ys.min() + 1)

    def contourArea(c):
        return float(len(c))

    cv2.resize, cv2.threshold, cv2.findContours = resize, threshold, findContours
    cv2.boundingRect, cv2.contourArea = boundingRect, contourArea
    return cv2


_installed = False


def install():
    """Make `import clip`, `import WeCLIP_model`, `import pytorch_grad_cam`, `import utils`
    resolve to the reference tree.  Idempotent."""
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError("reference tree not present")

    def ident(self, *a, **k):
        return self
    torch.Tensor.cuda = ident
    torch.nn.Module.cuda = ident

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _T:  # identity-callable transform
        def __init__(self, *a, **k):
            pass

        def __call__(self, x):
            return x
    tv = stub("torchvision")
    tvt = stub("torchvision.transforms", Compose=_T, Normalize=_T, ToTensor=_T, Resize=_T,
               CenterCrop=_T)
    tv.transforms = tvt
    stub("ftfy", fix_text=lambda s: s)
    lx = stub("lxml")
    lx.etree = stub("lxml.etree")
    stub("ttach")
    pd = stub("pydensecrf")
    pd.densecrf = stub("pydensecrf.densecrf")
    pd.utils = stub("pydensecrf.utils", unary_from_softmax=None, unary_from_labels=None)
    try:
        import imageio  # noqa: F401
    except Exception:
        stub("imageio")
    mm = stub("mmcv")
    mm.cnn = stub("mmcv.cnn", ConvModule=object)
    if "tqdm" not in sys.modules:
        try:
            import tqdm  # noqa: F401
        except Exception:
            stub("tqdm", tqdm=lambda x, *a, **k: x)
    sys.modules["cv2"] = _cv2_stub()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    _installed = True


def build_clip(state_dict):
    """reference build_model on CPU (clip/clip.py:146-149 does .float() for cpu)."""
    install()
    from clip.model import build_model
    m = build_model({k: v.clone() for k, v in state_dict.items()})
    return m.float().eval()
