#!/bin/bash
# PMC passes for one weight-gradient shape (args forwarded to wgrad_one.py).  Output summarised by pmc_sum.py
cd /tmp && export TMPDIR=/tmp
R=/root/repo
OUT=$R/gpurun_out/pmc_wgrad
rm -rf $OUT; mkdir -p $OUT
python $R/scripts/wgrad_one.py "$@"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVES SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python $R/scripts/wgrad_one.py "$@" > /dev/null 2>&1 || echo "pass $i failed"
done
python $R/scripts/pmc_sum.py $OUT
