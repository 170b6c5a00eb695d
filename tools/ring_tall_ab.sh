#!/bin/bash
# tall matrix waves of the ring kernel (RgCfg<..., MT = 4>: 4 waves x 128 pixels, default on 32-wide tiles) vs the 8-wave kernel (PWS_OPT_EXPERIMENT 105)
cd ${GRAFT_REPO_ROOT:-.}
export CONV_BENCH_STORE16=1 CONV_BENCH_ROTATE=3
for shape in "CONV_K3S1 64 128 128 128 128" "CONV_K3S1 64 256 256 64 64" "CONV_K3S1 64 128 128 256 128" "CONV_K3S1 64 64 64 256 256" "CONVT_K4S2 64 128 128 256 64" "CONVT_K4S2 64 64 64 512 128" "CONVT_K4S2 64 32 32 1024 256"; do
  for e in 105 0 105 0; do
    PWS_EXPERIMENT=$e python tools/conv_bench.py $shape bf16 2>&1 | grep -v amdgpu.ids
    PWS_EXPERIMENT=$e python tools/conv_bench.py $shape bf16 dgrad 2>&1 | grep -v amdgpu.ids
  done
done
