"""ctypes binding of libweclip_hip.so (the C ABI declared in include/weclip_hip.h).

PyTorch is used only as the owner of device memory and streams: tensors are handed to the
library as raw device pointers plus sizes, on torch's current HIP stream.  There is no
fallback path -- a missing library, a CPU tensor or a wrong dtype raises immediately.
"""
import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libweclip_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "weclip_hip.h")

_CT = {
    "int": ctypes.c_int, "float": ctypes.c_float, "long": ctypes.c_long,
    "int64_t": ctypes.c_int64, "double": ctypes.c_double,
}


def parse_header(path=HEADER_PATH):
    """[(name, restype, [(ctype, argname)])] for every prototype in the public header."""
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    protos = []
    for m in re.finditer(r"^\s*(const char\*|int|void)\s+(wc_\w+)\s*\(([^;]*?)\)\s*;", txt, flags=re.M | re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        alist = []
        if args.strip() not in ("", "void"):
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    alist.append((ctypes.c_void_p, a.split("*")[-1].strip()))
                else:
                    t, n = a.rsplit(" ", 1)
                    alist.append((_CT[t.replace("const ", "").strip()], n))
        protos.append((name, ret, alist))
    return protos


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it first (python weclip-vit-comer_amd/build.py or "
                "__graft_entry__.build()); this package has no non-HIP fallback")
        self.cdll = ctypes.CDLL(LIB_PATH)
        self.cdll.wc_last_error.restype = ctypes.c_char_p
        self._fn = {}
        for name, ret, args in parse_header():
            f = getattr(self.cdll, name)          # AttributeError if the .so lacks a declared symbol
            f.argtypes = [t for t, _ in args]
            f.restype = ctypes.c_char_p if ret == "const char*" else (None if ret == "void" else ctypes.c_int)
            self._fn[name] = f

    def __getattr__(self, name):
        fn = self.__dict__.get("_fn", {}).get(name)
        if fn is None:
            raise AttributeError(name)

        def call(*args):
            rc = fn(*args)
            if rc != 0:
                raise RuntimeError(f"{name}: {self.cdll.wc_last_error().decode()} (code {rc})")
        return call


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib


_gpu_ok = None


def require_gpu():
    global _gpu_ok
    if _gpu_ok is None:
        _gpu_ok = torch.cuda.is_available()
    if not _gpu_ok:
        raise RuntimeError("weclip_vit_comer_amd needs an AMD GPU (torch.cuda.is_available() is "
                           "False); there is no CPU path")


def ptr(t, dtype=None, name="tensor"):
    """Device pointer of a contiguous CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name}: expected a CUDA tensor")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name}: expected a contiguous tensor")
    return ctypes.c_void_p(t.data_ptr())


def stream():
    """Raw handle of torch's current HIP stream on the current device.  (torch.cuda.current_stream() builds
    a Python Stream object, ~9 us; this is called once per kernel launch, ~700 times per training step.)"""
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def int_array(values):
    return (ctypes.c_int * len(values))(*[int(v) for v in values])
