#!/bin/bash
# same-box A/B of the weight-gradient dispatch inside the full training step (variant 9: generic only, 0: default)
for v in ${1:-9 0 9 0}; do
  MDCV_WGRAD_VARIANT=$v python bench.py --workload ${WL:-both} --steps 30 --warmup 5 --no-cpu-baseline --no-breakdown 2>/dev/null > /tmp/ab.json
  python - "$v" <<'PY'
import json, sys
d = json.loads(open("/tmp/ab.json").read())
print("variant", sys.argv[1], "img/s %.1f" % d["value"], {k: v for k, v in d["config"].items() if "rekt" in k.lower()}, flush=True)
PY
done
