#!/bin/bash
# Profile artefacts of one round, produced on the GPU box:   gpurun -- bash scripts/profile_round.sh r02 [yolo|rektnet]
#   gpurun_out/prof_<tag>/<tag>_<wl>_bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `python bench.py --workload <wl>` (the command roofline is quoted on)
#   gpurun_out/prof_<tag>/<tag>_<wl>_bench_under_rocprof.json the JSON line that profiled run printed (its roofline_kernels must agree with the csv)
#   gpurun_out/prof_<tag>/<tag>[_rektnet]_pmc_hbm_traffic.json HBM bytes per kernel launch: separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes, FETCH_SIZE doubled
#                                                             (MI355X_MICROARCH.md, HBM section), stamped with the kernel fingerprint bench.py checks
#   gpurun_out/prof_<tag>/<tag>_pmc_mfma_busy.json            SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE per kernel (MFMA utilisation)
# Copy the files into profiles/ and commit them.  Counter passes carry --kernel-trace only (no sys/hip/hsa trace domains).
TAG=${1:-r02}
WL=${2:-yolo}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
T=$OUT/tmp_$WL; rm -rf $T; mkdir -p $T

# (the headline loop only: the reference-loop and classes=1 legs of the default line run other step counts of the same kernels)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $T/stats -- python $R/bench.py --workload $WL --no-cpu-baseline --no-fp32 --no-ref-loop --no-classes1 \
    > $OUT/${TAG}_${WL}_bench_under_rocprof.json 2> $T/stats_err.log || echo "stats pass failed"
f=$(ls $T/stats/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" $OUT/${TAG}_${WL}_bench_kernel_stats.csv

if [ "$WL" = "yolo" ] || [ "$WL" = "rektnet" ]; then
  CMD="python $R/bench.py --workload $WL --steps 4 --warmup 1 --no-cpu-baseline --no-breakdown --no-fp32 --no-ref-loop --no-classes1"
  export MDCV_WGRAD_STREAM=0      # counters are per dispatch but device-wide: one kernel at a time
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $T/fetch -- $CMD > /dev/null 2> $T/fetch_err.log || echo "fetch pass failed"
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $T/write -- $CMD > /dev/null 2> $T/write_err.log || echo "write pass failed"
  timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $T/mfma -- $CMD > /dev/null 2> $T/mfma_err.log || echo "mfma pass failed"
  unset MDCV_WGRAD_STREAM
  python $R/scripts/pmc_round.py $T $OUT $TAG $WL
fi
rm -rf $T/stats $T/fetch $T/write $T/mfma
ls -la $OUT
