"""List kernels whose global loads are waited for one at a time (`load; s_waitcnt vmcnt(0)` right behind it).

hipcc compiles a bounds branch around a load (`if (i < n) v = p[i];`, `v = ok ? p[i] : 0.f;`, `continue` on an outside tap) into
branch + load + `s_waitcnt vmcnt(0)`: the loads of a thread then run as a chain of dependent memory latencies instead of together.
A profile only shows a slow kernel; the disassembly shows the cause.  Usage: python tools/isa_scan.py [file.hip ...] (default: every
.hip of the package); compiles each with `hipcc -S --cuda-device-only` into /tmp/isa and prints, per kernel, the number of loads, the
number waited for alone and the load / wait / branch sequence (L load, D LDS-DMA, wN s_waitcnt vmcnt(N), | branch, S store, B barrier).
"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "weclip-vit-comer_amd", "csrc")


def disassemble(path, out_dir="/tmp/isa"):
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.abspath(path)
    out = os.path.join(out_dir, os.path.basename(path) + ".s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only",
                    "-I", os.path.dirname(path), path, "-o", out], check=True, stderr=subprocess.DEVNULL)
    return out


def kernels(asm):
    text = open(asm).read()
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M):
        yield m.group(1), [l.strip() for l in m.group(2).split("\n") if l.strip() and not l.strip().startswith(";")]


def scan(lines):
    loads = alone = 0
    seq = []
    for i, l in enumerate(lines):
        if re.match(r"(global|buffer)_load", l):
            if "lds" in l:
                seq.append("D")
                continue
            loads += 1
            seq.append("L")
            if any(lines[i + k].startswith("s_waitcnt vmcnt(0)") for k in (1, 2) if i + k < len(lines)):
                alone += 1
        elif l.startswith("s_waitcnt") and "vmcnt" in l:
            seq.append("w" + re.search(r"vmcnt\((\d+)\)", l).group(1))
        elif l.startswith("s_cbranch"):
            seq.append("|")
        elif re.match(r"(global|buffer)_store", l):
            seq.append("S")
        elif l.startswith("s_barrier"):
            seq.append("B")
    return loads, alone, "".join(seq)


def report(files, threshold=3):
    """[(loads waited for alone, loads, file, mangled kernel name, sequence)] of the kernels at or above the threshold."""
    rows = []
    for f in files:
        for name, lines in kernels(disassemble(f)):
            loads, alone, seq = scan(lines)
            if alone >= threshold:
                rows.append((alone, loads, os.path.basename(f), name, seq))
    return rows


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(SRC, "*.hip")))
    rows = report(files)
    for alone, loads, f, name, seq in sorted(rows, reverse=True):
        print(f"{alone:4d} of {loads:4d} loads waited for alone  {f:18s} {name[:90]}")
        print("      " + seq[:200])


if __name__ == "__main__":
    main()
