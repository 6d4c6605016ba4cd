set -e
R=$PWD
mkdir -p gpurun_out/r04
python -m pytest tests/test_backward_ops_gpu.py tests/test_comer_gpu.py tests/test_weclip_gpu.py tests/test_graph_step_gpu.py -x -q > gpurun_out/r04/gputest_29.log 2>&1 || { tail -40 gpurun_out/r04/gputest_29.log; exit 1; }
tail -2 gpurun_out/r04/gputest_29.log
bash tools/refresh_profiles.sh r04
cp gpurun_out/r04_traffic.json profiles/r04_traffic.json
python bench.py > gpurun_out/r04/bench_default_v3.json 2> gpurun_out/r04/bench_default_v3.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench_default_v3.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','repeat_ms_per_step')})
print(d['roofline']['frac'], d['roofline']['traffic'])
print('with_comer', d['with_comer']['ms_per_step'])
for x in d['with_comer']['roofline']:
    if 'bucket' in x['kernel'] or 'gather' in x['kernel'] or 'km' in x['kernel'] or 'ln_bwd' in x['kernel']: print('   ', x['kernel'], x['achieved'], x['unit'], x['avg_launch_us'])
PY
