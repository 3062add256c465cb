#!/bin/bash
# Phase timestamps (wall_clock64) of workgroup 0 of post_image_kernel: builds the library with -DMDCV_POST_TS, runs the
# post-processing bench (prints "post phase N: x us" on the 50th call), then rebuilds the shipped library.
# Run on a GPU box:  gpurun -- bash scripts/post_phases.sh     (needs hipcc there; otherwise build here first)
set -e
cd "$(dirname "$0")/../mit-driverless-cv-traininginfra_amd/csrc"
touch postprocess.hip
make CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I. -DMDCV_POST_TS" >/dev/null
(cd ../.. && python bench.py --workload postprocess --steps 200 --warmup 20 --no-cpu-baseline | grep phase)
touch postprocess.hip
make >/dev/null
