set -e
R=$PWD
mkdir -p gpurun_out/r04
python -m pytest tests/test_comer_gpu.py tests/test_comer_fullsize_gpu.py tests/test_primitives_gpu.py tests/test_weclip_gpu.py tests/test_graph_step_gpu.py tests/test_bench_size_golden_gpu.py -x -q > gpurun_out/r04/gputest_13.log 2>&1 || { tail -40 gpurun_out/r04/gputest_13.log; exit 1; }
tail -2 gpurun_out/r04/gputest_13.log
python tools/comer_bench.py > gpurun_out/r04/comer_bench_11.txt 2>&1; tail -1 gpurun_out/r04/comer_bench_11.txt
for w in 512 256 128; do
CB_HEAD_WGRAD_WGS=$w python bench.py --repeats 3 --no-cpu-baseline --no-extras > gpurun_out/r04/bench_6_$w.json 2> gpurun_out/r04/bench_6_$w.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r04/bench_6_$w.json').read().strip().splitlines()[-1])
print($w, {k:d[k] for k in ('value','ms_per_step','repeat_ms_per_step')}, d['roofline']['frac'])
PY
done
