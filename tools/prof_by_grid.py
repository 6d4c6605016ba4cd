"""Per (kernel, grid) summary of a rocprofv3 --kernel-trace results db: which launch shapes a kernel's time is in.
usage: prof_by_grid.py results.db steps [name-substring ...]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2])
pats = sys.argv[3:]
cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
if not cols:                                  # a view: take the names from a row
    cur = db.execute("select * from kernels limit 1")
    cols = [d[0] for d in cur.description]
gc = [c for c in cols if re.search(r"grid", c, re.I)]
wc = [c for c in cols if re.search(r"workgroup", c, re.I)]
sel = ", ".join(gc + wc)
rows = db.execute(f"select name, {sel}, count(*), sum(end-start), avg(end-start), min(end-start) from kernels "
                  f"group by name, {sel} order by sum(end-start) desc").fetchall()
print("kernel," + ",".join(gc + wc) + ",launches_per_step,ms_per_step,avg_us,min_us")
for r in rows:
    nm = re.sub(r"\(.*", "", r[0]).replace("void ", "")[:60]
    if pats and not any(p in nm for p in pats):
        continue
    n = len(gc) + len(wc)
    print(f"\"{nm}\"," + ",".join(str(v) for v in r[1:1 + n]) + f",{r[1 + n] / steps:.1f},{r[2 + n] / 1e6 / steps:.3f},{r[3 + n] / 1e3:.1f},{r[4 + n] / 1e3:.1f}")
