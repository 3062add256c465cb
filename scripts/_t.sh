python -m pytest -m gpu -q --timeout=600 tests/test_gpu_kernels.py -k "wgrad" 2>&1 | tail -3
echo "--- w4 256 blocks"; MDCV_WGRAD_VARIANT=30256 python scripts/wgrad_ab.py 0,9 all 2>&1 | grep -v amdgpu | tail -8
echo "--- w8 256 blocks"; MDCV_WGRAD_VARIANT=30256,30002 python scripts/wgrad_ab.py 0 yolo 2>&1 | grep -v amdgpu | head -3
echo "--- w4 128 blocks"; python scripts/wgrad_ab.py 0 yolo 2>&1 | grep -v amdgpu | head -3
run() { env $1 python bench.py --workload yolo --no-cpu-baseline --no-breakdown --no-fp32 --steps 20 --warmup 6 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-50s %8.1f img/s %7.3f ms' % ('$1', l['value'], l['ms_per_step']))"; }
run "MDCV_WGRAD_VARIANT=30002"
run "MDCV_X=w4_128"
run "MDCV_WGRAD_VARIANT=30160"
run "MDCV_WGRAD_VARIANT=30192"
run "MDCV_WGRAD_VARIANT=30256"
run "MDCV_WGRAD_VARIANT=30096"
