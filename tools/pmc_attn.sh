#!/bin/bash
# SQ counters of the attention kernels alone (tools/attn_one.py), separate --pmc passes.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_attn_$i -- python3 $R/tools/attn_one.py > $R/gpurun_out/pmc_attn_$i.log 2>&1 || echo "pass $i failed"
done
python3 $R/tools/pmc_counters.py $R/gpurun_out/pmc_attn_1 $R/gpurun_out/pmc_attn_2 $R/gpurun_out/pmc_attn_3 $R/gpurun_out/pmc_attn_4 $R/gpurun_out/pmc_attn_5 --match attn_ > $R/gpurun_out/pmc_attn.txt
find $R/gpurun_out -name "*.csv" -path "*pmc_attn_*" -size +2M -delete
echo done
