timeout 900 python -m pytest tests/test_gpu_emd.py -m gpu -x -q -k "sharp_level" 2>&1 | tail -5
