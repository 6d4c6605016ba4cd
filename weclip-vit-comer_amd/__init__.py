"""MI355X-native WeCLIP forward/CAM hot path.

Hand-written HIP kernels for gfx950 behind a C-ABI shared library (csrc/, include/weclip_hip.h)
and a Python host side that mirrors the reference's module layout:

    weclip_vit_comer_amd.clip              (model, myAtt, clip_tool, utils)
    weclip_vit_comer_amd.pytorch_grad_cam  (GradCAM)
    weclip_vit_comer_amd.WeCLIP_model      (PAR, segformer_head, Decoder.TransDecoder,
                                            model_attn_aff_voc, model_attn_aff_coco)

`install_dropin()` registers those sub-packages under the reference's top-level import names
(`clip`, `pytorch_grad_cam`, `WeCLIP_model`) so the reference's training / evaluation scripts
import this implementation unmodified.  There is no CPU fallback: every op raises if the HIP
library or a GPU is missing.
"""
import sys

__version__ = "0.1.0"


def install_dropin():
    """Make `import clip`, `import pytorch_grad_cam`, `import WeCLIP_model` resolve here."""
    import importlib
    for name in ("clip", "pytorch_grad_cam", "WeCLIP_model"):
        mod = importlib.import_module(f"{__name__}.{name}")
        sys.modules[name] = mod
        prefix = f"{__name__}.{name}."
        for k, v in list(sys.modules.items()):
            if k.startswith(prefix):
                sys.modules[name + "." + k[len(prefix):]] = v
