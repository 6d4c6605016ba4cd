set -e
R=$PWD
mkdir -p gpurun_out/r04
python -m pytest tests -m gpu -x -q > gpurun_out/r04/gputest_12.log 2>&1 || { tail -40 gpurun_out/r04/gputest_12.log; exit 1; }
tail -2 gpurun_out/r04/gputest_12.log
python tools/comer_bench.py > gpurun_out/r04/comer_bench_10.txt 2>&1; tail -1 gpurun_out/r04/comer_bench_10.txt
python bench.py --repeats 3 --no-cpu-baseline > gpurun_out/r04/bench_5.json 2> gpurun_out/r04/bench_5.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench_5.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','repeat_ms_per_step')})
print('roofline',d['roofline']['kernel'],d['roofline']['frac'])
for leg in ('with_comer','seg_trans_branch','exact_precision','fast_gemm_fp32_par','encoder_only_b32'):
    if leg in d: print(leg, d[leg].get('ms_per_step'))
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_base -o p -- python3 $R/bench.py --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline --no-extras --timer-stride 0 --roof-steps 0 --single-stream > $R/gpurun_out/r04/prof_base.log 2>&1
cd $R
python3 tools/prof_summary.py gpurun_out/prof_base/p_results.db 14 120 > gpurun_out/r04/base_step_kernel_stats_v1.csv
find gpurun_out -name "*.db" -delete
head -70 gpurun_out/r04/base_step_kernel_stats_v1.csv
