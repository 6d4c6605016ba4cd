"""MI355X-native WeCLIP forward/CAM hot path.

Hand-written HIP kernels for gfx950 behind a C-ABI shared library (csrc/, include/weclip_hip.h)
and a Python host side that mirrors the reference's module layout:

    weclip_vit_comer_amd.clip              (model, myAtt, clip_tool, utils)
    weclip_vit_comer_amd.pytorch_grad_cam  (GradCAM)
    weclip_vit_comer_amd.WeCLIP_model      (PAR, segformer_head, Decoder.TransDecoder,
                                            model_attn_aff_voc, model_attn_aff_coco)

`install_dropin()` installs an import hook that maps the reference's top-level import names
(`clip`, `pytorch_grad_cam`, `WeCLIP_model`, `utils`) and their sub-modules onto those sub-packages,
so the reference's training / evaluation scripts import this implementation unmodified.  There is no CPU fallback: every op raises if the HIP
library or a GPU is missing.
"""
import sys

__version__ = "0.2.0"


def register_torch_ops():
    """Register the hot-path operators with the PyTorch dispatcher (`torch.ops.weclip.*`, torch_ops.py)."""
    from . import torch_ops
    return torch_ops.OPS


DROPIN_NAMES = ("clip", "pytorch_grad_cam", "WeCLIP_model", "utils")


class _AliasLoader:
    """Loader that hands back the already-imported package module for an aliased name, so that
    `clip.model` and `weclip_vit_comer_amd.clip.model` are ONE module object (one set of classes,
    one HIP library handle)."""

    def __init__(self, real):
        self.real = real
        self.real_spec = getattr(real, "__spec__", None)

    def create_module(self, spec):
        return self.real

    def exec_module(self, module):
        # importlib rebinds module.__spec__ to the alias spec; put the real one back so relative imports
        # inside the module keep resolving against weclip_vit_comer_amd.*
        if self.real_spec is not None:
            module.__spec__ = self.real_spec


class _AliasFinder:
    """sys.meta_path finder: `import clip[.x]`, `pytorch_grad_cam[.x]`, `WeCLIP_model[.x]`, `utils[.x]`
    -> `weclip_vit_comer_amd.<same>`.  Names this package does not provide (e.g. the reference's
    `utils.imutils`) are left to the regular finders (see install_dropin(reference_root=...))."""

    def find_spec(self, fullname, path=None, target=None):
        import importlib
        import importlib.util
        if fullname.split(".", 1)[0] not in DROPIN_NAMES:
            return None
        real_name = f"{__name__}.{fullname}"
        try:
            if importlib.util.find_spec(real_name) is None:
                return None
        except (ImportError, AttributeError, ValueError):
            return None
        real = importlib.import_module(real_name)
        spec = importlib.util.spec_from_loader(fullname, _AliasLoader(real), is_package=hasattr(real, "__path__"))
        return spec


_finder = None


def install_dropin(reference_root=None):
    """Make the reference's import lines (scripts/dist_clip_voc.py:17-23, test_msc_flip_*.py) resolve to this
    package from a fresh interpreter:

        from WeCLIP_model.model_attn_aff_voc import WeCLIP;  from utils.losses import get_aff_loss
        from utils.camutils import cams_to_affinity_label;   from utils.optimizer import PolyWarmupAdamW
        from utils import evaluate;  import clip;  from pytorch_grad_cam import GradCAM

    `utils` resolves to this package's `utils` (losses, camutils, optimizer, evaluate, AverageMeter: the
    modules on the hot path or imported next to it by the training script).  `reference_root`, if given, is a
    checkout of the reference whose `utils/` directory is appended to that package's search path, so the helper
    modules this package does not provide (imutils, dcrf, ...) still import from the user's tree."""
    global _finder
    import importlib
    import os
    if _finder is None:
        _finder = _AliasFinder()
        sys.meta_path.insert(0, _finder)
    for name in DROPIN_NAMES:
        stale = sys.modules.get(name)
        real = importlib.import_module(f"{__name__}.{name}")
        if stale is not None and stale is not real:
            for k in [k for k in sys.modules if k == name or k.startswith(name + ".")]:
                del sys.modules[k]
        sys.modules[name] = real
        prefix = f"{__name__}.{name}."
        for k, v in list(sys.modules.items()):
            if k.startswith(prefix) and v is not None:
                sys.modules[name + "." + k[len(prefix):]] = v
    if reference_root is not None:
        extra = os.path.join(reference_root, "utils")
        up = importlib.import_module(f"{__name__}.utils").__path__
        if os.path.isdir(extra) and extra not in list(up):
            up.append(extra)
