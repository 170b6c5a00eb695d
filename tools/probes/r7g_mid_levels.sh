#!/bin/bash
# the <= 16x16 levels of the fp32 forward at the lockstep batch (16) through pws_conv2d_fwd: product selection (0), the Winograd ring forced (58), never the ring (50)
cd ${GRAFT_REPO_ROOT:-.}
for shape in "CONV_K3S1 16 16 16 256 256" "CONVT_K3S1 16 16 16 512 512" "CONVT_K4S2 16 16 16 1024 256" "CONV_K3S2 16 32 32 512 256" "CONV_K3S2 16 16 16 512 256" "CONV_K3S1 16 8 8 256 256" "CONVT_K3S1 16 8 8 512 512" "CONVT_K4S2 16 8 8 1024 256" "CONVT_K4S2 8 8 8 512 256" "CONV_K3S2 8 32 32 256 256"; do
  for e in 0 58 50; do PWS_EXPERIMENT=$e CONV_BENCH_ROTATE=4 python tools/conv_bench.py $shape wino 2>/dev/null | tail -1; done
done
