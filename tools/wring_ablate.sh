#!/bin/bash
# Timing-only ablations of the Winograd ring kernel (PWS_EXPERIMENT = 1000 + mask; see WringParams.ablate)
export CONV_BENCH_ROTATE=3
for e in ${WRING_EXPS:-0 1064 1096 1104 1112 1120 1124}; do
  export PWS_EXPERIMENT=$e
  python tools/conv_bench.py CONV_K3S1 8 64 64 256 256 wino
done 2>&1 | grep -v amdgpu.ids
