set -e
R=$PWD
mkdir -p gpurun_out/r04
python -m pytest tests/test_comer_gpu.py tests/test_comer_fullsize_gpu.py -x -q > gpurun_out/r04/gputest_15.log 2>&1 || { tail -40 gpurun_out/r04/gputest_15.log; exit 1; }
tail -2 gpurun_out/r04/gputest_15.log
python tools/comer_bench.py > gpurun_out/r04/comer_bench_12.txt 2>&1; tail -1 gpurun_out/r04/comer_bench_12.txt
python bench.py --repeats 1 --no-cpu-baseline > gpurun_out/r04/bench_8.json 2> gpurun_out/r04/bench_8.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench_8.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'])
print('with_comer', d['with_comer']['ms_per_step'])
for x in d['with_comer']['roofline']: print('   ', x)
PY
