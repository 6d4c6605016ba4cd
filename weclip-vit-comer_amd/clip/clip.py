"""`clip.load` for local checkpoints (reference clip/clip.py:95-149, file-path branch only: the
name branch downloads from the internet and is not supported)."""
import os
from collections import OrderedDict

import torch

from .model import build_model

__all__ = ["load", "build_model"]


def load(name, device="cuda", jit=False, download_root=None):
    """`name`: path to a TorchScript archive or a state-dict file, or an in-memory state dict.
    Returns (model.eval() on `device`, None) -- the second slot is the preprocessing transform
    in the reference, which the WeCLIP path never uses."""
    if isinstance(name, dict):
        sd = name
    elif isinstance(name, str) and os.path.isfile(name):
        try:
            sd = torch.jit.load(name, map_location="cpu").eval().state_dict()
        except RuntimeError:
            raw = torch.load(name, map_location="cpu")
            sd = OrderedDict((k.replace("module.", ""), v) for k, v in raw.items())
    else:
        raise RuntimeError(f"Model {name} not found: pass a local checkpoint path (no download here)")
    return build_model(sd).to(device), None
