#!/bin/bash
# the dominant YOLOv3 layer shapes through conv_one.py (B H Cin Cout k stride mode variant iters)
cd /root/repo
for sh in "32 52 128 256 3 1 0" "32 52 128 256 3 1 1" "32 26 256 512 3 1 0" "32 26 256 512 3 1 1" "32 13 512 1024 3 1 0" "32 13 512 1024 3 1 1" \
          "32 104 64 128 3 1 0" "32 52 256 128 1 1 0" "32 26 512 256 1 1 0" "32 208 32 64 3 1 0" "32 416 32 64 3 2 0"; do
  python scripts/conv_one.py $sh ${1:--1} 60 | sed 's/shape//'
done
