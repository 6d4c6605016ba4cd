#!/bin/bash
# Regenerate the judged rocprofv3 artefacts of `bench.py` on the GPU box (run through gpurun from the repo root):
#   kernel-trace stats + two separate PMC passes (FETCH_SIZE, WRITE_SIZE), as the MI355X guide prescribes.
# The bench command runs 2 set-up steps (eager + graph capture: the capture executes nothing), W warm-up, K timed and
# 5 idle-queue steps: 2 + W + K + 5 executed steps (the capturing step replays its graph right away); --repeats 1, or the
# timed region runs five times and the per-step normalisation below is off by that factor.
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
# --single-stream: the head forward on the step's own stream, so that the trace times every kernel alone (the timed bench
# steps run it on a second stream beside the CAM chain: 12.9 vs 13.2 ms per step)
ARGS="--steps 5 --warmup 2 --repeats 1 --no-cpu-baseline --no-extras --timer-stride 0 --roof-steps 0 --single-stream"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats -o p -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-extras --timer-stride 0 --roof-steps 0 --single-stream > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-extras --timer-stride 0 --roof-steps 0 --single-stream > $R/gpurun_out/pmc_write.log 2>&1
cd $R
python3 tools/prof_summary.py gpurun_out/prof_stats/p_results.db 14 100 > gpurun_out/${TAG}_kernel_stats.csv
python3 tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/${TAG}
# the same two PMC passes with the ViT-CoMer inserts (BASELINE configs[2] as written) -> ${TAG}_comer_traffic.json
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_c -- python3 $R/bench.py --comer --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-extras --timer-stride 0 --roof-steps 0 --single-stream > $R/gpurun_out/pmc_fetch_c.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write_c -- python3 $R/bench.py --comer --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-extras --timer-stride 0 --roof-steps 0 --single-stream > $R/gpurun_out/pmc_write_c.log 2>&1
cd $R
python3 tools/pmc_traffic.py gpurun_out/pmc_fetch_c gpurun_out/pmc_write_c gpurun_out/${TAG}_comer
find gpurun_out -name "*.db" -delete                     # gpurun copies back at most 64 MiB
find gpurun_out -name "*.csv" -size +1M -delete
echo done
