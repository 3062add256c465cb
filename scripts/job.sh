mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv or shift or dgrad" 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_gpu_models.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python scripts/ab_step.py "c-30,P0;c-32,P1;c-32,P0;c-30,P1" 4 40 > gpurun_out/r4_step1.txt 2>&1
tail -4 gpurun_out/r4_step1.txt
