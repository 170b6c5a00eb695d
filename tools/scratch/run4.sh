timeout 600 python -m pytest tests/test_hip_bf16.py tests/test_hip_backward.py -x -q -m gpu 2>&1 | tail -4
for e in 0 2; do
PWS_EXPERIMENT=$e timeout 120 python tools/wgrad_bench.py CONV_K5S1 32 256 256 32 64 2>&1 | tail -1
done
timeout 600 python tools/train_profile.py --batch 32 --math bf16 2>&1 | tail -28 | head -16
