set -e
R=$PWD
mkdir -p gpurun_out/r04
python -m pytest tests/test_primitives_gpu.py tests/test_comer_gpu.py tests/test_torch_ops_gpu.py tests/test_weclip_gpu.py -x -q > gpurun_out/r04/gputest_19.log 2>&1 || { tail -40 gpurun_out/r04/gputest_19.log; exit 1; }
tail -2 gpurun_out/r04/gputest_19.log
python tools/wgrad_bench.py > gpurun_out/r04/wgrad_bench_1.txt 2>&1; grep "TF/s" gpurun_out/r04/wgrad_bench_1.txt
python tools/gemm_row_bench.py 2>&1 | grep "gelu\|M=86016 N=256 K=256 f32 out + resid   " > gpurun_out/r04/gemm_row_bench_5.txt; cat gpurun_out/r04/gemm_row_bench_5.txt
python tools/comer_bench.py > gpurun_out/r04/comer_bench_14.txt 2>&1; tail -1 gpurun_out/r04/comer_bench_14.txt
python -m pytest tests/test_comer_fullsize_gpu.py -x -q > gpurun_out/r04/gputest_19b.log 2>&1 || { tail -40 gpurun_out/r04/gputest_19b.log; exit 1; }
tail -2 gpurun_out/r04/gputest_19b.log
python bench.py --repeats 3 --no-cpu-baseline > gpurun_out/r04/bench_9.json 2> gpurun_out/r04/bench_9.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench_9.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','repeat_ms_per_step')}, d['roofline']['frac'])
for leg in ('with_comer','seg_trans_branch','exact_precision','fast_gemm_fp32_par','encoder_only_b32'):
    if leg in d: print(leg, d[leg].get('ms_per_step'))
PY
