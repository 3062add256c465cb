#!/usr/bin/env python3
"""Soak of the 2-rank data-parallel spawn that tests/test_gpu_dp.py uses (one hang was seen on a fresh box and papered over with a
retry).  Runs the mini-Darknet 2-rank exchange N times WITHOUT the retry; a worker that is still alive after 200 s dumps its Python stacks
(faulthandler), which this script prints -- the blocking call by name.   usage: dp_spawn_soak.py [runs] [which=yolo|rektnet|auto_uneven]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_dp as T  # noqa: E402

if __name__ == "__main__":
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    which = sys.argv[2] if len(sys.argv) > 2 else "yolo"
    times, hangs = [], 0
    for i in range(runs):
        t0 = time.time()
        try:
            T._run(2, which, attempts=1)
        except AssertionError as e:
            hangs += 1
            print(f"run {i}: HANG / failure after {time.time() - t0:.1f} s\n{e}", flush=True)
            continue
        times.append(time.time() - t0)
        if i % 10 == 9:
            print(f"run {i + 1}/{runs}: {hangs} hangs so far, last {times[-1]:.1f} s, max {max(times):.1f} s", flush=True)
    print(f"{runs} runs of the 2-rank '{which}' spawn: {hangs} hangs; wall per run min {min(times):.1f} / median {sorted(times)[len(times) // 2]:.1f} / max {max(times):.1f} s")
