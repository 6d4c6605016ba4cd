set -e
R=$PWD
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_primitives_gpu.py -x -q -k "gemm_row" > gpurun_out/r04/gputest_row.log 2>&1 || { tail -40 gpurun_out/r04/gputest_row.log; exit 1; }
tail -2 gpurun_out/r04/gputest_row.log
timeout -k 10 300 python tools/gemm_row_bench.py > gpurun_out/r04/gemm_row_bench_2.txt 2>&1 || { tail -20 gpurun_out/r04/gemm_row_bench_2.txt; exit 1; }
grep "M=86016 N=256 K=256\|M=16384 N=256 K=256" gpurun_out/r04/gemm_row_bench_2.txt
python tools/comer_bench.py > gpurun_out/r04/comer_bench_2.txt 2>&1
tail -1 gpurun_out/r04/comer_bench_2.txt
python -m pytest tests/test_torch_ops_gpu.py tests/test_comer_fullsize_gpu.py tests/test_comer_gpu.py tests/test_weclip_gpu.py -q > gpurun_out/r04/gputest_5.log 2>&1 || true
tail -4 gpurun_out/r04/gputest_5.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_comer -o p -- python3 $R/bench.py --comer --steps 5 --warmup 2 --no-cpu-baseline --no-extras --timer-stride 0 --roof-steps 0 --single-stream > $R/gpurun_out/r04/prof_comer.log 2>&1
cd $R
python3 tools/prof_summary.py gpurun_out/prof_comer/p_results.db 14 120 > gpurun_out/r04/comer_step_kernel_stats_v1.csv
find gpurun_out -name "*.db" -delete
head -50 gpurun_out/r04/comer_step_kernel_stats_v1.csv
