run() { env $1 python bench.py --workload yolo --no-cpu-baseline --no-fp32 --steps 20 --warmup 6 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-30s %8.1f img/s %7.3f ms  pack %s' % ('$1', l['value'], l['ms_per_step'], l['workloads']['yolo']['kernel_ms_per_step'].get('mdcv_pack_weights_batched')))"; }
run "MDCV_X=1"
run "MDCV_X=2"
python -m pytest -m gpu -q --timeout=600 tests/test_gpu_models.py -k "mini or roundtrip or pipelined" 2>&1 | tail -2
