#!/bin/bash
# same-box A/B of MDCV_BN_FUSE_SKIP masks (which data-gradient classes keep the stand-alone BatchNorm-backward reduce pass)
mkdir -p gpurun_out/fp
for rep in 1 2 3; do
  for p in ${MASKS:-0 13 1 12 29 45}; do
    MDCV_BN_FUSE_SKIP=$p python bench.py --workload yolo --no-cpu-baseline --no-breakdown --steps 40 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('skip-mask $p rep $rep', round(d['value'], 1))" | tee -a gpurun_out/fp/ab.txt
  done
done
