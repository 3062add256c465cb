#!/bin/bash
# 3x3 stride-1 layer shapes for the shift kernel at several weight-ring depths (variant -3..-6) vs the im2col kernel (variant 11 / 9)
cd /root/repo
for v in -3 -4; do
for sh in "32 52 128 256 3 1 0" "32 52 128 256 3 1 1" "32 26 256 512 3 1 0" "32 26 256 512 3 1 1" "32 13 512 1024 3 1 0" "32 13 512 1024 3 1 1" "32 104 64 128 3 1 0"; do
  python scripts/conv_one.py $sh $v 60 2>/dev/null | sed 's/shape//'
done
done
