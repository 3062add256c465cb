python -m pytest -m gpu -q --timeout=900 tests/test_gpu_kernels.py tests/test_gpu_robust.py 2>&1 | tail -6
run() { env $1 python bench.py --workload $2 --no-cpu-baseline --no-breakdown --no-fp32 --steps 20 --warmup 6 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-60s %-8s %8.1f img/s %7.3f ms' % ('$1', '$2', l['value'], l['ms_per_step']))"; }
run "MDCV_WGRAD_STREAM=0 MDCV_WGRAD_BATCH=1" yolo
run "MDCV_WGRAD_STREAM=0 MDCV_WGRAD_BATCH=4" yolo
run "MDCV_WGRAD_STREAM=0 MDCV_WGRAD_BATCH=4 MDCV_WGRAD_VARIANT=30256" yolo
run "MDCV_WGRAD_BATCH=1" rektnet
run "MDCV_WGRAD_VARIANT=1800" rektnet
run "MDCV_WGRAD_VARIANT=30256" rektnet
run "MDCV_WGRAD_VARIANT=30064" rektnet
