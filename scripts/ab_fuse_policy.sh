#!/bin/bash
# Same-box A/B of which data gradients carry the fused BatchNorm-backward sums (engine.Plan._fuse_pays, MDCV_BN_FUSE_SKIP bit mask:
# 1: 52^2 3x3, 2: 26^2 3x3, 4: 104^2 1x1, 8: stride-2 52->104, 16: 52^2 1x1, 32: 104^2 3x3, 64: 26^2 1x1; set bit = stand-alone reduce pass).
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { MDCV_BN_FUSE_SKIP=$1 python $R/bench.py --workload yolo --no-cpu-baseline --no-breakdown --no-fp32 --steps 20 --warmup 6 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('skip %-4s %8.1f img/s %7.3f ms' % ('$1', l['value'], l['ms_per_step']))"; }
for m in ${@:-13 29 77 93 15 12 5 9 45 127 0 13}; do run $m; done
