timeout 600 python -m pytest tests/test_hip_bf16.py tests/test_hip_backward.py -x -q -m gpu 2>&1 | tail -4
for cfg in "CONV_K3S1 32 256 256 64 64" "CONV_K3S1 32 64 64 256 256" "CONV_K3S1 32 16 16 512 512" "CONV_K3S2 32 256 256 64 128" "CONVT_K4S2 32 128 128 128 64" "CONV_K3S1 32 128 128 128 128"; do
for e in 0 1; do
PWS_EXPERIMENT=$e timeout 120 python tools/wgrad_bench.py $cfg 2>&1 | tail -1
done
done
timeout 600 python tools/train_profile.py --batch 32 --math bf16 2>&1 | tail -28
