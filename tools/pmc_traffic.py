"""Condense two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as the MI355X guide
prescribes) of `bench.py` into per-kernel HBM traffic per launch.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ...
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01

Units/corrections (guides/MI355X_MICROARCH.md §HBM): both counters are in KiB; on gfx950 FETCH_SIZE
reports exactly half of the bytes of a coalesced streaming read, so reads = 2 * FETCH_SIZE (checked here
on par_iter_kernel: 2*FETCH = 112.6 MB vs 107 MB algorithmic reads per launch; WRITE_SIZE matches the
6.29 MB / 100.7 MB written by par_iter / par_affinity exactly)."""
import collections
import csv
import glob
import json
import sys


def load(d, counter):
    f = glob.glob(f"{d}/*/*_counter_collection.csv")[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return acc


def main(fetch_dir, write_dir, out_prefix):
    fe, wr = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    rows, summary = [], {}
    for k in sorted(fe, key=lambda k: -sum(fe[k])):
        f, w = fe[k], wr.get(k, [0.0])
        fa, wa = sum(f) / len(f), sum(w) / len(w)
        traffic = (2.0 * fa + wa) * 1024.0
        rows.append((k, len(f), fa, wa, traffic))
        summary[k] = {"launches": len(f), "fetch_size_kib_avg": fa, "write_size_kib_avg": wa,
                      "hbm_bytes_per_launch": traffic}
    with open(out_prefix + "_pmc_traffic.csv", "w") as fh:
        fh.write("kernel,launches,FETCH_SIZE_KiB_avg,WRITE_SIZE_KiB_avg,hbm_bytes_per_launch(2*FETCH+WRITE)\n")
        for r in rows:
            fh.write("%s,%d,%.2f,%.2f,%.0f\n" % r)
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("_b", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                     "weclip-vit-comer_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    summary["__meta__"] = {"source_hash": b.source_hash(), "note": "hash of csrc/*.hip, *.h + hipcc flags at collection time"}
    json.dump(summary, open(out_prefix + "_traffic.json", "w"), indent=1)
    print("wrote", out_prefix + "_pmc_traffic.csv", "and _traffic.json;", len(rows), "kernels")


if __name__ == "__main__":
    main(*sys.argv[1:4])
