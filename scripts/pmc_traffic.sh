#!/bin/bash
# HBM traffic per kernel from separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the YOLOv3 bench command.
# Run on the GPU box:  gpurun -- bash scripts/pmc_traffic.sh   -> gpurun_out/pmc_traffic.json (copy to profiles/)
cd /tmp && export TMPDIR=/tmp
R=/root/repo
OUT=$R/gpurun_out/pmc_traffic
rm -rf $OUT; mkdir -p $OUT
CMD="python $R/bench.py --workload yolo --steps 2 --warmup 1 --no-cpu-baseline --no-breakdown"
export MDCV_WGRAD_STREAM=0
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- $CMD > /dev/null 2>&1 || echo "fetch pass failed"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- $CMD > /dev/null 2>&1 || echo "write pass failed"
python $R/scripts/pmc_traffic.py $OUT $R/gpurun_out/pmc_traffic.json
rm -rf $OUT/fetch/*/*kernel_trace.csv $OUT/write/*/*kernel_trace.csv
