set -x
timeout 900 python -m pytest tests/test_hip_bf16.py tests/test_hip_backward.py tests/test_hip_netg.py tests/test_hip_perceptual.py -x -q -m gpu 2>&1 | tail -15
for cfg in "CONV_K3S1 32 256 256 64 64" "CONV_K3S1 32 128 128 128 128" "CONV_K3S1 32 64 64 256 256" "CONV_K3S2 32 256 256 64 128" "CONVT_K4S2 32 128 128 128 64"; do
CONV_BENCH_STORE16=1 CONV_BENCH_ROTATE=3 timeout 300 python tools/conv_bench.py $cfg bf16 2>&1 | tail -3
done
timeout 600 python tools/train_profile.py --batch 32 --math bf16 2>&1 | tail -30
