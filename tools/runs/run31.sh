set -e
R=$PWD
mkdir -p gpurun_out/r04
bash tools/refresh_profiles.sh r04
cp gpurun_out/r04_traffic.json profiles/r04_traffic.json
cp gpurun_out/r04_comer_traffic.json profiles/r04_comer_traffic.json
python bench.py > gpurun_out/r04/bench_default_v4.json 2> gpurun_out/r04/bench_default_v4.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench_default_v4.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','repeat_ms_per_step')})
print(d['roofline']['frac'], d['roofline']['traffic'])
print('with_comer', d['with_comer']['ms_per_step'], d['with_comer']['traffic_source'][:80])
for x in d['with_comer']['roofline']: print('   ', x['kernel'], x['achieved'], x['unit'], x['avg_launch_us'], x['traffic'])
PY
