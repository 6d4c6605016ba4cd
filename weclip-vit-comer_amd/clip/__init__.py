"""HIP-backed counterpart of the reference `clip` package (model, myAtt, clip_tool, utils)."""
from .clip import load, build_model  # noqa: F401
