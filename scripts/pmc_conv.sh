#!/bin/bash
# PMC passes for one conv shape (args forwarded to conv_one.py).  Output: gpurun_out/pmc_one/*.csv summarised by pmc_sum.py
cd /tmp && export TMPDIR=/tmp
R=/root/repo
OUT=$R/gpurun_out/pmc_one
rm -rf $OUT; mkdir -p $OUT
python $R/scripts/conv_one.py "$@"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python $R/scripts/conv_one.py "$@" > /dev/null 2>&1 || echo "pass $i failed"
done
python $R/scripts/pmc_sum.py $OUT
