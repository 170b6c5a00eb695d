#!/bin/bash
# Winograd ring kernel (PWS_EXPERIMENT 0) vs the kernels that run without it (50) on the generator's conv_same and transposed
# layers (batch 8).  WRING_EXPS selects other variants (timing-only ablations: 1000 + mask, see WringParams.ablate).
export CONV_BENCH_ROTATE=3
for e in ${WRING_EXPS:-0 50}; do
  export PWS_EXPERIMENT=$e
  python tools/conv_bench.py CONV_K3S1 8 128 128 128 128 wino
  python tools/conv_bench.py CONV_K3S1 8 64 64 256 256 wino
  python tools/conv_bench.py CONV_K3S1 8 32 32 512 512 wino
  python tools/conv_bench.py CONVT_K4S2 8 128 128 256 64 wino
  python tools/conv_bench.py CONVT_K4S2 8 64 64 512 64 wino
  python tools/conv_bench.py CONVT_K4S2 8 32 32 1024 128 wino
done 2>&1 | grep -v amdgpu.ids
