#!/bin/bash
# Which launches is a training step's time sensitive to?  Re-times the step with one family of launches dropped from the lists
# (MDCV_ABLATE, results are wrong by construction: timing only).  The difference to the full step is what removing / hiding that
# family could buy at most.   usage: ablate.sh [yolo|rektnet]
R=${GRAFT_REPO_ROOT:-/root/repo}
WL=${1:-yolo}
run() { MDCV_ABLATE="$1" python $R/bench.py --workload $WL --no-cpu-baseline --no-breakdown --no-fp32 --steps 20 --warmup 6 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-60s %8.1f img/s %7.3f ms' % ('$1' or 'full', l['value'], l['ms_per_step']))"; }
run ""
run "conv2d_wgrad"
run "mdcv_bn_act_fwd"
run "mdcv_bn_act_bwd_apply"
run "mdcv_bn_stats_finalize,mdcv_bn_bwd_finalize_rows"
run "mdcv_bn_act_bwd_reduce_finalize"
run "mdcv_pack_weights_batched"
run "mdcv_conv2d_dgrad_bnsums"
run "conv2d_wgrad,mdcv_bn_act_fwd,mdcv_bn_act_bwd_apply,mdcv_bn_stats_finalize,mdcv_bn_bwd_finalize_rows,mdcv_bn_act_bwd_reduce_finalize"
run ""
