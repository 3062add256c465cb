timeout 1200 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "pw_block" -s 2>&1 | grep -E "Mismatch|ACTUAL|DESIRED|Max |x:|y:|^pw plans|passed|failed" | cut -c1-400 | head -12
