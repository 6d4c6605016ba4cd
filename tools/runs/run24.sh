set -e
mkdir -p gpurun_out/r04
timeout -k 10 300 python -m pytest tests/test_primitives_gpu.py tests/test_torch_ops_gpu.py -x -q > gpurun_out/r04/gputest_24.log 2>&1 || { tail -40 gpurun_out/r04/gputest_24.log; exit 1; }
tail -2 gpurun_out/r04/gputest_24.log
timeout -k 10 200 python tools/wgrad_bench.py > gpurun_out/r04/wgrad_bench_4.txt 2>&1; grep "TF/s" gpurun_out/r04/wgrad_bench_4.txt
timeout -k 10 300 python -m pytest tests/test_weclip_gpu.py tests/test_comer_gpu.py -x -q > gpurun_out/r04/gputest_24b.log 2>&1 || { tail -40 gpurun_out/r04/gputest_24b.log; exit 1; }
tail -2 gpurun_out/r04/gputest_24b.log
python tools/comer_bench.py > gpurun_out/r04/comer_bench_17.txt 2>&1; echo "$(tail -1 gpurun_out/r04/comer_bench_17.txt)"
python bench.py --repeats 3 --no-cpu-baseline --no-extras > gpurun_out/r04/bench_11.json 2> gpurun_out/r04/bench_11.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench_11.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','repeat_ms_per_step')}, d['roofline']['frac'])
PY
sed -i 's/^_WGRAD_WGS = 512/_WGRAD_WGS = 256/' weclip-vit-comer_amd/head_engine.py
python bench.py --repeats 3 --no-cpu-baseline --no-extras > gpurun_out/r04/bench_11b.json 2> gpurun_out/r04/bench_11b.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench_11b.json').read().strip().splitlines()[-1])
print('head 256:', {k:d[k] for k in ('value','ms_per_step','repeat_ms_per_step')}, d['roofline']['frac'])
PY
