#!/bin/bash
# Everything profiles/ holds for one round, in one GPU call:   gpurun -- bash scripts/round_evidence.sh r02
#   full `pytest -m gpu`, the rocprofv3 / PMC sets of scripts/profile_round.sh for both workloads, the default and the joint bench lines,
#   the per-geometry conv launch tables.  Output under gpurun_out/prof_<tag>/; copy into profiles/ and commit.
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $R
python -m pytest tests -m gpu -q 2>&1 | tail -3 > $OUT/${TAG}_pytest_gpu_tail.txt
bash scripts/profile_round.sh $TAG yolo > /dev/null 2>&1
bash scripts/profile_round.sh $TAG rektnet > /dev/null 2>&1
cd $R
cp $OUT/${TAG}_*.csv $OUT/${TAG}_pmc_*.json $OUT/${TAG}_*_under_rocprof.json $R/profiles/ 2>/dev/null   # bench.py quotes traffic / *_in_step from profiles/ when the fingerprint matches
python bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/bench_default.err
cp $R/gpurun_out/bench_detail.json $OUT/${TAG}_bench_detail.json 2>/dev/null    # per-kernel tables of that run (the stdout line stays < 3.5 KB)
python bench.py --workload joint > $OUT/${TAG}_joint_bench.json 2> $OUT/joint.err
python bench.py --workload yolo --no-cpu-baseline --no-fp32 --no-ref-loop --no-classes1 --dump-launches $OUT/yolo_launches.json > /dev/null 2>&1 && python scripts/layer_table.py $OUT/yolo_launches.json > $OUT/${TAG}_yolo_conv_launch_table.txt
python bench.py --workload rektnet --no-cpu-baseline --no-fp32 --no-ref-loop --dump-launches $OUT/rektnet_launches.json > /dev/null 2>&1 && python scripts/layer_table.py $OUT/rektnet_launches.json > $OUT/${TAG}_rektnet_conv_launch_table.txt
rm -f $OUT/*_launches.json
# one steady-state step as a two-queue timeline
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tl && mkdir -p /tmp/tl && \
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python $R/bench.py --workload yolo --steps 20 --warmup 10 --no-cpu-baseline --no-breakdown --no-fp32 --no-ref-loop --no-classes1 > /dev/null 2>&1 ; \
  f=$(find /tmp/tl -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python $R/scripts/step_timeline.py $f 2 > $OUT/${TAG}_yolo_step_timeline.txt 2>&1 )
cat $OUT/${TAG}_pytest_gpu_tail.txt
ls -la $OUT
