timeout 1500 python -m pytest tests/test_hip_bf16.py tests/test_hip_perceptual.py tests/test_hip_backward.py -x -q -m gpu 2>&1 | tail -8
timeout 300 python tools/vgg_profile.py --batch 64 2>&1 | tail -6
