set -e
R=$PWD
mkdir -p gpurun_out/r04
bash tools/refresh_profiles.sh r04
mkdir -p profiles
cp gpurun_out/r04_traffic.json profiles/r04_traffic.json
python bench.py > gpurun_out/r04/bench_default_v2.json 2> gpurun_out/r04/bench_default_v2.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench_default_v2.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','repeat_ms_per_step')})
print(d['roofline'])
print('with_comer', d['with_comer']['ms_per_step'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['best_threads'])
PY
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
python -m pytest tests -m gpu -x -q > gpurun_out/r04/gputest_28.log 2>&1 || { tail -40 gpurun_out/r04/gputest_28.log; exit 1; }
tail -2 gpurun_out/r04/gputest_28.log
