#!/bin/bash
# same-box A/B of two builds of libmdcv_hip.so (.ab/libold.so vs .ab/libnew.so) inside the training step
# usage: ab_lib.sh [rounds] ; WL=yolo|rektnet|both
P=mit-driverless-cv-traininginfra_amd
for i in $(seq ${1:-2}); do
  for w in old new; do
    cp .ab/lib$w.so $P/libmdcv_hip.so
    python bench.py --workload ${WL:-both} --steps 30 --warmup 5 --no-cpu-baseline --no-breakdown 2>/dev/null > /tmp/ab.json
    python - "$w" <<'PY'
import json, sys
d = json.loads(open("/tmp/ab.json").read())
print(sys.argv[1], " ".join("%s %.1f" % (k, v["images_per_sec"]) for k, v in d["workloads"].items()), flush=True)
PY
  done
done
cp .ab/libnew.so $P/libmdcv_hip.so
