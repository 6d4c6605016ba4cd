set -e
R=$PWD
mkdir -p gpurun_out/r04
python -m pytest tests/test_comer_gpu.py tests/test_comer_fullsize_gpu.py -x -q > gpurun_out/r04/gputest_10.log 2>&1 || { tail -40 gpurun_out/r04/gputest_10.log; exit 1; }
tail -2 gpurun_out/r04/gputest_10.log
for w in 512 256 1024; do
CB_WGRAD_WGS=$w python tools/comer_bench.py > gpurun_out/r04/comer_bench_8_$w.txt 2>&1; echo "wgs $w: $(tail -1 gpurun_out/r04/comer_bench_8_$w.txt)"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_comer -o p -- python3 $R/bench.py --comer --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline --no-extras --timer-stride 0 --roof-steps 0 --single-stream > $R/gpurun_out/r04/prof_comer.log 2>&1
cd $R
python3 tools/prof_summary.py gpurun_out/prof_comer/p_results.db 14 120 > gpurun_out/r04/comer_step_kernel_stats_v3.csv
find gpurun_out -name "*.db" -delete
head -50 gpurun_out/r04/comer_step_kernel_stats_v3.csv
