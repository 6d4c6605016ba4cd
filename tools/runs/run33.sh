set -e
R=$PWD
mkdir -p gpurun_out/r04
python -m pytest tests -m gpu -x -q > gpurun_out/r04/gputest_33.log 2>&1 || { tail -40 gpurun_out/r04/gputest_33.log; exit 1; }
tail -2 gpurun_out/r04/gputest_33.log
bash tools/refresh_profiles.sh r04
cp gpurun_out/r04_traffic.json profiles/r04_traffic.json
cp gpurun_out/r04_comer_traffic.json profiles/r04_comer_traffic.json
python bench.py > gpurun_out/r04/bench_default_v5.json 2> gpurun_out/r04/bench_default_v5.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench_default_v5.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','repeat_ms_per_step')})
print(d['roofline']['frac'], d['roofline']['traffic'])
print('with_comer', d['with_comer']['ms_per_step'])
for x in d['with_comer']['roofline']:
    if 'bucket' in x['kernel']: print('   ', x['kernel'], x['avg_launch_us'])
PY
grep "layernorm_kernel<4>\|sum_slices_wb_multi" gpurun_out/r04_kernel_stats.csv
