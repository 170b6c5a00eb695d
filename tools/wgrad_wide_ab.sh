#!/bin/bash
# transposed-layer weight gradients of configs[2] (batch 64): wgrad_bf16_kernel (80) / 64 x 64 ring pairs (86) / 128 x 64 ring pairs (81, round 4)
cd ${GRAFT_REPO_ROOT:-.}
for shape in "64 128 128 256 64" "64 128 128 128 64" "64 64 64 512 64" "64 64 64 256 64" "64 32 32 1024 128" "64 32 32 512 128" "64 16 16 1024 256" "64 16 16 512 256"; do
  for e in 80 86 81 84 0; do PWS_EXPERIMENT=$e python tools/wgrad_bench.py CONVT_K4S2 $shape 2>&1 | grep wgrad; done
done
