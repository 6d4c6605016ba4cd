set -e
R=$PWD
mkdir -p gpurun_out/r04
python -m pytest tests/test_comer_gpu.py tests/test_backward_ops_gpu.py tests/test_primitives_gpu.py -x -q > gpurun_out/r04/gputest_8.log 2>&1 || { tail -40 gpurun_out/r04/gputest_8.log; exit 1; }
tail -2 gpurun_out/r04/gputest_8.log
python tools/comer_bench.py > gpurun_out/r04/comer_bench_6.txt 2>&1; tail -1 gpurun_out/r04/comer_bench_6.txt
python -m pytest tests/test_comer_fullsize_gpu.py tests/test_graph_step_gpu.py tests/test_weclip_gpu.py tests/test_bench_size_golden_gpu.py -q > gpurun_out/r04/gputest_9.log 2>&1 || true
tail -4 gpurun_out/r04/gputest_9.log
python bench.py --comer --steps 10 --warmup 3 --no-extras --no-cpu-baseline --roof-steps 4 --repeats 2 > gpurun_out/r04/bench_comer_4.json 2> gpurun_out/r04/bench_comer_4.err || tail -20 gpurun_out/r04/bench_comer_4.err
python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r04/bench_3.json 2> gpurun_out/r04/bench_3.err || tail -20 gpurun_out/r04/bench_3.err
python - <<'PY'
import json
for f in ('bench_comer_4','bench_3'):
    d=json.loads([l for l in open(f'gpurun_out/r04/{f}.json') if l.startswith('{')][-1])
    print(f, {k:d[k] for k in ('value','ms_per_step','repeat_ms_per_step')})
    for r in [d['roofline']]+d['roofline_other'][:16]: print('   ', r['kernel'], r['achieved'], r['unit'], r['frac'], r['avg_launch_us'], r['share_of_eager_step'])
PY
