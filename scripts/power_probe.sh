#!/bin/bash
# sample clocks / power while the YOLOv3 bench loop runs
python bench.py --workload yolo --steps 1500 --warmup 20 --no-breakdown --no-cpu-baseline --no-fp32 > /tmp/bench_out.txt 2>&1 &
BP=$!
for i in $(seq 1 80); do
  if ! kill -0 $BP 2>/dev/null; then break; fi
  echo -n "t=$i "
  /opt/rocm/bin/rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|Power \(W\)|GPU use" | sed 's/GPU\[0\]\t\t: //' | tr '\n' ' '
  echo
  sleep 0.5
done
wait $BP
tail -c 300 /tmp/bench_out.txt
