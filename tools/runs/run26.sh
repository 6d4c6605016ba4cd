set -e
R=$PWD
mkdir -p gpurun_out/r04
bash tools/refresh_profiles.sh r04
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_comer -o p -- python3 $R/bench.py --comer --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline --no-extras --timer-stride 0 --roof-steps 0 --single-stream > $R/gpurun_out/r04/prof_comer.log 2>&1
cd $R
python3 tools/prof_summary.py gpurun_out/prof_comer/p_results.db 14 120 > gpurun_out/r04/comer_step_kernel_stats_v6.csv
find gpurun_out -name "*.db" -delete
head -30 gpurun_out/r04/comer_step_kernel_stats_v6.csv
