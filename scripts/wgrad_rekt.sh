#!/bin/bash
# isolated weight-gradient time of RektNet's 3x3 layers (batch 256, 80x80) and YOLOv3's low-channel layers, variants $1 (default "0 9")
for v in ${1:-0 9}; do
  echo "== variant $v"
  for sh in "256 80 16 16 3 1" "256 80 16 16 3 2" "256 80 16 32 3 2" "256 80 32 32 3 1" "256 80 32 64 3 2" "256 80 64 64 3 1" "256 80 64 128 3 2" \
            "32 208 32 64 3 1" "32 104 64 128 3 1"; do
    python scripts/wgrad_one.py $sh $v 50 2>/dev/null
  done
done
