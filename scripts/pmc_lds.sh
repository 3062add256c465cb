#!/bin/bash
# LDS bank-conflict counters for one launch loop.  usage: pmc_lds.sh <script.py> args...
cd /tmp && export TMPDIR=/tmp
R=/root/repo
OUT=$R/gpurun_out/pmc_lds
rm -rf $OUT; mkdir -p $OUT
S=$1; shift
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --kernel-trace --output-format csv -d $OUT/p1 -- python $R/scripts/$S "$@" > /dev/null 2>&1 || echo "pass failed"
python $R/scripts/pmc_sum.py $OUT | grep -v "reduce\|pack"
