set -e
R=$PWD
mkdir -p gpurun_out/r04
python bench.py --repeats 1 --no-cpu-baseline > gpurun_out/r04/bench_7.json 2> gpurun_out/r04/bench_7.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench_7.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'])
print('with_comer', d['with_comer']['ms_per_step'])
for x in d['with_comer']['roofline']: print('   ', x['kernel'], x.get('achieved'), x.get('unit'), x.get('frac'), x.get('avg_us'), x.get('share'))
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_comer -o p -- python3 $R/bench.py --comer --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline --no-extras --timer-stride 0 --roof-steps 0 --single-stream > $R/gpurun_out/r04/prof_comer.log 2>&1
cd $R
python3 tools/prof_summary.py gpurun_out/prof_comer/p_results.db 14 120 > gpurun_out/r04/comer_step_kernel_stats_v5.csv
find gpurun_out -name "*.db" -delete
head -45 gpurun_out/r04/comer_step_kernel_stats_v5.csv
