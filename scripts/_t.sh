run() { env $1 python bench.py --workload $2 --no-cpu-baseline --no-breakdown --no-fp32 --steps 20 --warmup 6 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-60s %-8s %8.1f img/s %7.3f ms' % ('$1', '$2', l['value'], l['ms_per_step']))"; }
run "MDCV_BN_FUSE_MAXROWS=4096" rektnet
run "MDCV_X=1" rektnet
run "MDCV_BN_FUSE_MAXROWS=4096" yolo
run "MDCV_X=1" yolo
python -m pytest -m gpu -q --timeout=900 -x tests/test_gpu_kernels.py tests/test_gpu_models.py tests/test_gpu_redzone.py 2>&1 | tail -4
