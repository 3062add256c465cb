mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_models.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/pw_models.txt
cat gpurun_out/pw_models.txt
timeout 900 python scripts/ab_step.py "P0;P1" 4 40 > gpurun_out/pw_step.txt 2>&1
tail -3 gpurun_out/pw_step.txt
