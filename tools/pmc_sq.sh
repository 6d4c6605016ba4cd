#!/bin/bash
# SQ counters (MFMA busy cycles, LDS bank conflicts, wave cycles) of the GEMM kernels (tools/gemm_pmc.py), the attention
# kernels (tools/attn_one.py) and the PAR sweep (tools/par_bench.py): separate --pmc passes, summarised per kernel.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
n=0
for prog in gemm_pmc.py attn_one.py par_bench.py; do
  for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; do
    n=$((n+1))
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_sq_$n -- python3 $R/tools/$prog > $R/gpurun_out/pmc_sq_$n.log 2>&1 || echo "pass $n failed"
  done
done
python3 $R/tools/pmc_counters.py $(ls -d $R/gpurun_out/pmc_sq_*/) > $R/gpurun_out/pmc_sq_all.txt
grep -v "^ " $R/gpurun_out/pmc_sq_all.txt | wc -l
find $R/gpurun_out -path "*pmc_sq_*" -name "*.csv" -size +1M -delete
echo done
