for e in 0 1 2 4 3 6; do
PWS_EXPERIMENT=$e timeout 120 python tools/wgrad_bench.py CONV_K3S1 32 256 256 64 64 2>&1 | tail -1
done
for e in 0 1 2 4; do
PWS_EXPERIMENT=$e timeout 120 python tools/wgrad_bench.py CONV_K3S1 32 64 64 256 256 2>&1 | tail -1
done
for e in 0 1 2 4; do
PWS_EXPERIMENT=$e timeout 120 python tools/wgrad_bench.py CONV_K3S1 32 16 16 512 512 2>&1 | tail -1
done
