#!/bin/bash
# same-box A/B of environment settings over whole training steps: ab_env.sh <workload> "<env A>" "<env B>" ...   ("-" = no setting)
wl=$1; shift
mkdir -p gpurun_out/ab
for rep in 1 2 3; do
  for e in "$@"; do
    ee=$e; [ "$e" = "-" ] && ee=""
    env $ee python bench.py --workload $wl --no-cpu-baseline --no-breakdown --steps 40 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$e] rep $rep', ' '.join('%s %.1f' % (k, v['images_per_sec']) for k, v in d['workloads'].items()))" | tee -a gpurun_out/ab/ab.txt
  done
done
