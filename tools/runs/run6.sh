set -e
R=$PWD
mkdir -p gpurun_out/r04
python tools/row_probe.py > gpurun_out/r04/row_probe.txt 2>&1; cat gpurun_out/r04/row_probe.txt
python -m pytest tests/test_comer_fullsize_gpu.py tests/test_comer_gpu.py tests/test_backward_ops_gpu.py tests/test_graph_step_gpu.py -q > gpurun_out/r04/gputest_7.log 2>&1 || true
tail -5 gpurun_out/r04/gputest_7.log
python tools/comer_bench.py > gpurun_out/r04/comer_bench_4.txt 2>&1; tail -1 gpurun_out/r04/comer_bench_4.txt
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
  n=$((n+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_row_$n -- python3 $R/tools/row_probe.py 2 > $R/gpurun_out/pmc_row_$n.log 2>&1 || echo "pass $n failed"
done
cd $R
python3 tools/pmc_counters.py $(ls -d gpurun_out/pmc_row_*/) --match gemm_row > gpurun_out/r04/pmc_row.txt
cat gpurun_out/r04/pmc_row.txt
find gpurun_out -path "*pmc_row_*" -name "*.csv" -size +1M -delete
