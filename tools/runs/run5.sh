set -e
R=$PWD
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_primitives_gpu.py tests/test_backward_ops_gpu.py -x -q > gpurun_out/r04/gputest_row.log 2>&1 || { tail -40 gpurun_out/r04/gputest_row.log; exit 1; }
tail -2 gpurun_out/r04/gputest_row.log
timeout -k 10 300 python tools/gemm_row_bench.py > gpurun_out/r04/gemm_row_bench_3.txt 2>&1 || { tail -20 gpurun_out/r04/gemm_row_bench_3.txt; exit 1; }
grep "M=86016 N=256 K=256\|M=16384 N=256 K=256" gpurun_out/r04/gemm_row_bench_3.txt
python tools/comer_bench.py > gpurun_out/r04/comer_bench_3.txt 2>&1
tail -1 gpurun_out/r04/comer_bench_3.txt
python -m pytest tests -m gpu -q > gpurun_out/r04/gputest_6.log 2>&1 || true
tail -6 gpurun_out/r04/gputest_6.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04/bench_2.json 2> gpurun_out/r04/bench_2.err || tail -20 gpurun_out/r04/bench_2.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04/bench_2.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step','repeat_ms_per_step')})
print('roofline', d['roofline']['kernel'], d['roofline']['frac'])
for r in d['roofline_other'][:12]: print('   ', r['kernel'], r['achieved'], r['unit'], r['frac'], r['avg_launch_us'], r['share_of_eager_step'])
print('with_comer', d['with_comer']['ms_per_step'], [ (r['kernel'], r['achieved'], r['unit'], r['frac'], r.get('mfma_tflops'), r['avg_launch_us']) for r in d['with_comer'].get('roofline',[])])
for k in ('seg_trans_branch','exact_precision','fast_gemm_fp32_par','encoder_only_b32'): print(k, d[k]['ms_per_step'])
PY
