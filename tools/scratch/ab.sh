timeout 1200 python -m pytest tests/test_hip_bf16.py tests/test_hip_backward.py -x -q -m gpu 2>&1 | tail -2
for cfg in "CONVT_K4S2 32 128 128 128 64" "CONVT_K4S2 32 64 64 256 128" "CONVT_K4S2 32 32 32 512 256"; do
for e in 9 0 9 0; do
PWS_EXPERIMENT=$e timeout 300 python tools/wgrad_bench.py $cfg 2>&1 | tail -1
done
done
