#!/bin/bash
# ring weight gradients: previous tile / ring shapes (PWS_EXPERIMENT=88) against shorter tiles in a ring of 4 (0)
cd ${GRAFT_REPO_ROOT:-.}
for e in 88 0; do
  for shape in "64 128 128 256 64" "64 128 128 128 64" "64 64 64 512 64" "64 64 64 256 64" "64 32 32 1024 128" "64 32 32 512 128"; do PWS_EXPERIMENT=$e python tools/wgrad_bench.py CONVT_K4S2 $shape 2>&1 | grep wgrad; done
  for shape in "64 128 128 128 128" "64 64 64 256 256" "64 64 64 128 128" "64 32 32 256 256" "64 32 32 512 256"; do PWS_EXPERIMENT=$e python tools/wgrad_bench.py CONV_K3S1 $shape 2>&1 | grep wgrad; done
done
