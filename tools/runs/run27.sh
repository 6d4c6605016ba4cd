set -e
mkdir -p gpurun_out/r04
python -m pytest tests/test_backward_ops_gpu.py tests/test_comer_gpu.py tests/test_comer_fullsize_gpu.py -x -q > gpurun_out/r04/gputest_27.log 2>&1 || { tail -40 gpurun_out/r04/gputest_27.log; exit 1; }
tail -2 gpurun_out/r04/gputest_27.log
grep "CoMer inserts at" gpurun_out/r04/gputest_27.log || true
python tools/comer_bench.py > gpurun_out/r04/comer_bench_18.txt 2>&1; echo "$(tail -1 gpurun_out/r04/comer_bench_18.txt)"
python bench.py --comer --repeats 3 --no-cpu-baseline --no-extras > gpurun_out/r04/bench_12.json 2> gpurun_out/r04/bench_12.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench_12.json').read().strip().splitlines()[-1])
print('with comer:', {k:d[k] for k in ('value','ms_per_step','repeat_ms_per_step')}, d['roofline']['frac'])
PY
