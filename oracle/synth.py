"""Synthetic weights / inputs (TEST INFRASTRUCTURE view): the generators live in the package
(weclip-vit-comer_amd/synth.py) so that bench.py's measured path imports nothing from oracle/; the tests,
the golden-fixture generator and the CPU-baseline leg keep importing them from here."""
import importlib.util
import os
import sys

_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "weclip-vit-comer_amd", "synth.py")
_spec = importlib.util.spec_from_file_location("_weclip_synth", _path)
_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mod)
for _k, _v in vars(_mod).items():
    if not _k.startswith("__"):
        globals()[_k] = _v
