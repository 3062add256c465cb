#!/bin/bash
# Everything profiles/ holds for a round, in one GPU call:  gpurun --timeout 2400 -- bash scripts/final_profile.sh
# outputs under gpurun_out/final/ (copy to profiles/ with the round prefix)
R=/root/repo
OUT=$R/gpurun_out/final
rm -rf $OUT; mkdir -p $OUT
cd $R
bash scripts/pmc_traffic.sh > $OUT/pmc_traffic.log 2>&1
cp gpurun_out/pmc_traffic.json $OUT/pmc_hbm_traffic_yolo.json
cp gpurun_out/pmc_traffic.json profiles/r01_pmc_hbm_traffic_yolo_v2.json      # bench.py reads roofline.traffic from here
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --workload joint --no-cpu-baseline > $OUT/joint_bench.json 2>> $OUT/bench.err
python bench.py --workload postprocess > $OUT/postprocess_bench.json 2>> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $R/bench.py --no-cpu-baseline > /dev/null 2>&1 || echo "stats pass failed"
f=$(ls $OUT/stats/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" $OUT/bench_kernel_stats.csv
rm -rf $OUT/stats
cd $R; cut -c1-600 $OUT/bench.json
