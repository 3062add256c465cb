#!/bin/bash
# rocprofv3 kernel statistics of the YOLOv3-only bench command (the one roofline.achieved is quoted on) -> gpurun_out/ys/
R=/root/repo
OUT=$R/gpurun_out/ys
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $R/bench.py --workload yolo --no-cpu-baseline > $OUT/bench.json 2> $OUT/err.log || echo "stats pass failed"
f=$(ls $OUT/stats/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" $OUT/yolo_kernel_stats.csv
rm -rf $OUT/stats
