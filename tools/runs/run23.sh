set -e
mkdir -p gpurun_out/r04
python -m pytest tests/test_primitives_gpu.py tests/test_torch_ops_gpu.py -x -q > gpurun_out/r04/gputest_23.log 2>&1 || { tail -40 gpurun_out/r04/gputest_23.log; exit 1; }
tail -2 gpurun_out/r04/gputest_23.log
python tools/wgrad_bench.py > gpurun_out/r04/wgrad_bench_3.txt 2>&1; grep "TF/s" gpurun_out/r04/wgrad_bench_3.txt
python tools/comer_bench.py > gpurun_out/r04/comer_bench_16_256.txt 2>&1; echo "256: $(tail -1 gpurun_out/r04/comer_bench_16_256.txt)"
sed -i 's/^_WGRAD_WGS = 256/_WGRAD_WGS = 512/' weclip-vit-comer_amd/comer_engine.py
python tools/comer_bench.py > gpurun_out/r04/comer_bench_16_512.txt 2>&1; echo "512: $(tail -1 gpurun_out/r04/comer_bench_16_512.txt)"
python bench.py --repeats 3 --no-cpu-baseline --no-extras > gpurun_out/r04/bench_10.json 2> gpurun_out/r04/bench_10.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench_10.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','repeat_ms_per_step')}, d['roofline']['frac'])
PY
