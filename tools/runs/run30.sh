set -e
R=$PWD
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --repeats 1 --no-cpu-baseline --no-extras --timer-stride 0 --roof-steps 0 --single-stream"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats -o p -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_stats.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_comer -o p -- python3 $R/bench.py --comer $ARGS > $R/gpurun_out/r04/prof_comer.log 2>&1
cd $R
python3 tools/prof_summary.py gpurun_out/prof_stats/p_results.db 14 120 > gpurun_out/r04/base_step_kernel_stats_final.csv
python3 tools/prof_summary.py gpurun_out/prof_comer/p_results.db 14 120 > gpurun_out/r04/comer_step_kernel_stats_final.csv
find gpurun_out -name "*.db" -delete
head -8 gpurun_out/r04/base_step_kernel_stats_final.csv; head -8 gpurun_out/r04/comer_step_kernel_stats_final.csv
bash tools/pmc_sq.sh > gpurun_out/r04/pmc_sq.log 2>&1 || true
cp gpurun_out/pmc_sq_all.txt gpurun_out/r04/pmc_sq_counters.txt 2>/dev/null || true
wc -l gpurun_out/r04/pmc_sq_counters.txt
