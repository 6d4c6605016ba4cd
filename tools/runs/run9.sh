set -e
R=$PWD
mkdir -p gpurun_out/r04
python -m pytest tests/test_comer_gpu.py tests/test_weclip_gpu.py -x -q > gpurun_out/r04/gputest_8.log 2>&1 || { tail -40 gpurun_out/r04/gputest_8.log; exit 1; }
tail -2 gpurun_out/r04/gputest_8.log
python tools/comer_bench.py > gpurun_out/r04/comer_bench_7.txt 2>&1; tail -1 gpurun_out/r04/comer_bench_7.txt
CB_STEM=1 python tools/comer_bench.py > gpurun_out/r04/comer_bench_7s.txt 2>&1; tail -1 gpurun_out/r04/comer_bench_7s.txt
python -m pytest tests/test_comer_fullsize_gpu.py tests/test_graph_step_gpu.py -q > gpurun_out/r04/gputest_9.log 2>&1 || true
tail -4 gpurun_out/r04/gputest_9.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_comer -o p -- python3 $R/bench.py --comer --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline --no-extras --timer-stride 0 --roof-steps 0 --single-stream > $R/gpurun_out/r04/prof_comer.log 2>&1
cd $R
python3 tools/prof_summary.py gpurun_out/prof_comer/p_results.db 14 120 > gpurun_out/r04/comer_step_kernel_stats_v2.csv
find gpurun_out -name "*.db" -delete
head -64 gpurun_out/r04/comer_step_kernel_stats_v2.csv
