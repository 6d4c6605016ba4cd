set -e
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_primitives_gpu.py -x -q -k "gemm_row" > gpurun_out/r04/gputest_row.log 2>&1 || { tail -40 gpurun_out/r04/gputest_row.log; exit 1; }
tail -3 gpurun_out/r04/gputest_row.log
timeout -k 10 300 python tools/gemm_row_bench.py > gpurun_out/r04/gemm_row_bench_1.txt 2>&1 || { tail -20 gpurun_out/r04/gemm_row_bench_1.txt; exit 1; }
grep "M=86016 N=256 K=256\|M=16384 N=256 K=256" gpurun_out/r04/gemm_row_bench_1.txt
python tools/comer_bench.py > gpurun_out/r04/comer_bench_1.txt 2>&1
tail -1 gpurun_out/r04/comer_bench_1.txt
python -m pytest tests/test_torch_ops_gpu.py tests/test_comer_fullsize_gpu.py tests/test_comer_gpu.py tests/test_graph_step_gpu.py tests/test_affinity_gpu.py -q -s > gpurun_out/r04/gputest_3.log 2>&1 || true
tail -8 gpurun_out/r04/gputest_3.log
python -m pytest tests/test_weclip_gpu.py tests/test_bench_size_golden_gpu.py -q -s > gpurun_out/r04/gputest_4.log 2>&1 || true
grep "^\[\|passed\|failed" gpurun_out/r04/gputest_4.log | tail -20
python bench.py --steps 20 --warmup 5 > gpurun_out/r04/bench_1.json 2> gpurun_out/r04/bench_1.err || tail -20 gpurun_out/r04/bench_1.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04/bench_1.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step','repeat_ms_per_step')})
print('roofline', d['roofline']['kernel'], d['roofline']['frac'])
print('with_comer', d['with_comer']['ms_per_step'], [ (r['kernel'], r['achieved'], r['unit'], r['frac'], r.get('mfma_tflops')) for r in d['with_comer'].get('roofline',[])])
print('cpu', d.get('cpu_baseline'))
PY
