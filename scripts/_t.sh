for v in 16 17 16 17; do MDCV_CONV_VARIANT=$v python bench.py --workload yolo --no-cpu-baseline --no-breakdown --no-fp32 --steps 20 --warmup 6 2>/dev/null | python -c "
import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant $v', round(l['value'],1), round(l['ms_per_step'],3))"; done
python -m pytest -m gpu -q --timeout=600 tests/test_gpu_kernels.py -k "conv or dgrad" 2>&1 | tail -2
