echo "--- LA6 256"; MDCV_WGRAD_VARIANT=30256 python scripts/wgrad_ab.py 0 yolo 2>&1 | grep -v amdgpu.ids | head -3
