"""CPU checks of the C-ABI boundary: the library builds/loads without a GPU and exports every
symbol include/weclip_hip.h declares; argument errors are reported, not crashed on."""
import ctypes
import os
import subprocess

import pytest

import weclip_vit_comer_amd  # noqa: F401  (import alias for the hyphenated package dir)
from weclip_vit_comer_amd import _lib


@pytest.fixture(scope="module")
def so():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return ctypes.CDLL(_lib.LIB_PATH)


def test_header_symbols_are_exported(so):
    protos = _lib.parse_header()
    assert len(protos) >= 9
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    for name, _, _ in protos:
        assert name in exported, f"{name} declared in include/weclip_hip.h but not exported"
        getattr(so, name)
    # nothing undeclared leaks out of the ABI (wc_set_error is the internal error sink)
    assert {e for e in exported if e.startswith("wc_")} - {p[0] for p in protos} <= {"wc_set_error"}


def test_binding_loads_and_reports_argument_errors():
    lib = _lib.lib()
    assert lib.cdll.wc_version() >= 100
    with pytest.raises(RuntimeError, match="bad argument"):
        lib.wc_par_forward(None, None, None, None, None, 0, 0, 0, 0, _lib.int_array([1]), 1, 1, 1, None)
    with pytest.raises(RuntimeError, match="dilations"):
        lib.wc_par_affinity(ctypes.c_void_p(16), ctypes.c_void_p(16), 1, 4, 4,
                            _lib.int_array(range(1, 10)), 9, None)


def test_no_cpu_fallback():
    import torch
    from weclip_vit_comer_amd.WeCLIP_model.PAR import PAR
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="GPU"):
        PAR([1, 2], 2)(torch.zeros(1, 3, 8, 8), torch.zeros(1, 2, 8, 8))
