set -e
R=$PWD
mkdir -p gpurun_out/r04
python -m pytest tests/test_comer_gpu.py tests/test_backward_ops_gpu.py -x -q > gpurun_out/r04/gputest_8.log 2>&1 || { tail -40 gpurun_out/r04/gputest_8.log; exit 1; }
tail -2 gpurun_out/r04/gputest_8.log
python tools/comer_bench.py > gpurun_out/r04/comer_bench_5.txt 2>&1; tail -1 gpurun_out/r04/comer_bench_5.txt
python -m pytest tests/test_comer_fullsize_gpu.py tests/test_graph_step_gpu.py tests/test_primitives_gpu.py -q > gpurun_out/r04/gputest_9.log 2>&1 || true
tail -4 gpurun_out/r04/gputest_9.log
python bench.py --comer --steps 10 --warmup 3 --no-extras --no-cpu-baseline --roof-steps 4 --repeats 2 > gpurun_out/r04/bench_comer_3.json 2> gpurun_out/r04/bench_comer_3.err || tail -20 gpurun_out/r04/bench_comer_3.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04/bench_comer_3.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step','repeat_ms_per_step')})
for r in [d['roofline']]+d['roofline_other'][:24]: print('   ', r['kernel'], r['achieved'], r['unit'], r['frac'], r['avg_launch_us'], r['share_of_eager_step'])
PY
