set -e
R=$PWD
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp
for dir in c v; do for d in 0 1 2 3 4; do
MSDA_DBG=$d rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bk -o p -- python3 $R/tools/msda_bucket_probe.py $dir > $R/gpurun_out/prof_bk.log 2>&1
python3 - <<PY
import sqlite3,glob
db=glob.glob("$R/gpurun_out/prof_bk/*results.db")[0]
c=sqlite3.connect(db)
rows=c.execute("select name, avg(end-start), count(*) from kernels where name like '%msda_bucket%' or name like '%msda_gather%' group by name").fetchall()
print("$dir dbg=$d", [(r[0][:24], round(r[1]/1000,1), r[2]) for r in rows])
PY
rm -rf $R/gpurun_out/prof_bk
done; done > $R/gpurun_out/r04/bucket_probe.txt 2>&1
cat $R/gpurun_out/r04/bucket_probe.txt
