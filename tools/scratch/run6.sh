for cfg in "CONV_K3S1 32 256 256 64 64" "CONV_K3S1 32 256 256 128 64" "CONV_K3S1 32 128 128 128 128" "CONV_K3S1 32 64 64 256 256"; do
for e in 0 4 5 6; do
PWS_EXPERIMENT=$e CONV_BENCH_STORE16=1 CONV_BENCH_ROTATE=3 timeout 300 python tools/conv_bench.py $cfg bf16 2>&1 | tail -1
done
done
