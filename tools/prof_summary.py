"""Per-kernel summary (ms per step) of a rocprofv3 --kernel-trace results db."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 7
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows = db.execute("select name, count(*), sum(end-start), avg(end-start) from kernels group by name order by 3 desc").fetchall()
print("kernel,launches_per_step,ms_per_step,avg_us")
print(f"TOTAL,{sum(r[1] for r in rows)/steps:.1f},{sum(r[2] for r in rows)/1e6/steps:.3f},")
for r in rows[:top]:
    nm = re.sub(r"\(.*", "", r[0]).replace("void ", "").replace("at::native::", "")[:70]
    print(f"\"{nm}\",{r[1]/steps:.1f},{r[2]/1e6/steps:.3f},{r[3]/1e3:.1f}")
