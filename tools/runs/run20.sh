set -e
R=$PWD
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp
n=0
for sl in 64 128; do
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_ANY"; do
n=$((n+1))
WG_SLICES=$sl rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_km_$n -- python3 $R/tools/wgrad_pmc.py > $R/gpurun_out/pmc_km_$n.log 2>&1 || echo "pass $n failed"
done
python3 $R/tools/pmc_counters.py $(ls -d $R/gpurun_out/pmc_km_*/) --match gemm_km > $R/gpurun_out/r04/pmc_km_$sl.txt
rm -rf $R/gpurun_out/pmc_km_*
done
cd $R
cat gpurun_out/r04/pmc_km_64.txt gpurun_out/r04/pmc_km_128.txt
