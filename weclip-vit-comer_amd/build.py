"""Build libweclip_hip.so (hand-written HIP kernels + C-ABI) for gfx950 with hipcc.

In-tree build: objects under csrc/_build/, library next to this file.  hipcc cross-compiles
without a GPU, so this runs in the CPU-only build container as well as on the MI355X box.
    python weclip-vit-comer_amd/build.py [--force]
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_build")
LIB = os.path.join(HERE, "libweclip_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off",
         "-Wno-unused-result", "-I", CSRC] + os.environ.get("WECLIP_HIPCC_FLAGS", "").split()      # extra flags: experiments


# kernels whose hand-counted `s_waitcnt vmcnt(n)` waits are only correct without register spills (a spill's scratch access is
# one more entry in the in-order vmcnt queue): the build fails if the compiler reports scratch use for them
NO_SCRATCH = ("gemm_row.hip",)


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stamp(src):
    h = hashlib.sha1()
    # (include path normalised as in source_hash: objects built in one checkout stay valid when the tree is copied elsewhere)
    h.update(" ".join("<csrc>" if f == CSRC else f for f in FLAGS).encode())
    for f in [src] + sorted(f for f in os.listdir(CSRC) if f.endswith(".h")):
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def source_hash():
    """Content hash of every kernel source + header + the compile flags: identifies the build a profile was taken from
    (bench.py accepts a committed PMC traffic file only when its recorded hash equals this one)."""
    h = hashlib.sha1()
    # (the include path is absolute and differs between checkouts -- /root/repo here, a scratch directory on a GPU box --: the
    #  hash must not, or a traffic file collected on one box never matches on the next)
    h.update(" ".join("<csrc>" if f == CSRC else f for f in FLAGS).encode())
    for f in sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))):
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode())
            h.update(fh.read())
    return h.hexdigest()[:16]


def _compile(src):
    obj = os.path.join(OBJ, src[:-4] + ".o")
    stamp_file = obj + ".stamp"
    stamp = _stamp(src)
    if os.path.exists(obj) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return obj, False
    cmd = [HIPCC] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
    if src in NO_SCRATCH:
        cmd.append("-Rpass-analysis=kernel-resource-usage")
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    if src in NO_SCRATCH:
        import re
        spills = [int(v) for v in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", r.stderr)]
        if not spills or any(spills):
            raise RuntimeError(f"{src}: a kernel spills registers (scratch bytes per lane: {spills}); its counted "
                               "`s_waitcnt vmcnt(n)` waits assume that no scratch load / store is in the vmcnt queue")
    with open(stamp_file, "w") as fh:
        fh.write(stamp)
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 2)) as ex:
        res = list(ex.map(_compile, _sources()))
    objs = [o for o, _ in res]
    if any(c for _, c in res) or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[build] linked {LIB} from {len(objs)} objects", file=sys.stderr)
    elif verbose:
        print(f"[build] {LIB} up to date", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
