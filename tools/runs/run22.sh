set -e
mkdir -p gpurun_out/r04
for d in 0 1 2 3; do echo "== KM_DBG=$d"; KM_DBG=$d python tools/wgrad_bench.py 2>&1 | grep "M= 86016 N= 256 K= 256"; done > gpurun_out/r04/wgrad_dbg.txt
cat gpurun_out/r04/wgrad_dbg.txt
