#!/bin/bash
# Regenerate the judged rocprofv3 artefacts of `bench.py` on the GPU box (run through gpurun from the repo root):
#   kernel-trace stats + two separate PMC passes (FETCH_SIZE, WRITE_SIZE), as the MI355X guide prescribes.
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats -o p -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --timer-stride 0 > $R/gpurun_out/prof_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --timer-stride 0 > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --timer-stride 0 > $R/gpurun_out/pmc_write.log 2>&1
cd $R
python3 tools/prof_summary.py gpurun_out/prof_stats/p_results.db 7 100 > gpurun_out/kernel_stats.csv
python3 tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/r01
find gpurun_out -name "*.db" -delete                     # gpurun copies back at most 64 MiB
find gpurun_out -name "*.csv" -size +1M -delete
echo done
