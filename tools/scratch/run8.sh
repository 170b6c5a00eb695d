timeout 900 python -m pytest tests/test_hip_perceptual.py -x -q -m gpu -s 2>&1 | grep -v "^$" | grep -i "relative L2\|passed\|failed\|assert" | head
