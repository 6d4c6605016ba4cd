"""Average of every collected PMC counter per kernel from rocprofv3 --pmc output directories:
    python tools/pmc_counters.py <dir> [<dir> ...] [--match substring]"""
import collections, csv, glob, sys

dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
match = sys.argv[sys.argv.index("--match") + 1] if "--match" in sys.argv else ""
dirs = [d for d in dirs if d != match]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    for f in glob.glob(f"{d}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if match in k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"    {c:34s} {sum(v) / len(v):16.1f}  (n={len(v)})")
