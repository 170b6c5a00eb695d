#!/bin/bash
# stride-2 layers' weight gradients of configs[2] (batch 64): baseline library (LIBS[0]) against the build
cd ${GRAFT_REPO_ROOT:-.}
for shape in "64 256 256 64 64" "64 128 128 64 128" "64 128 128 128 128" "64 64 64 128 256" "64 64 64 256 256" "64 32 32 256 256" "64 32 32 512 256" "64 16 16 512 256"; do
  for lib in tools/_bin/ab_base.so pwstablenet_amd/libpwstable_hip.so; do echo -n "$(basename $lib) "; PWS_LIB_PATH=$PWD/$lib python tools/wgrad_bench.py CONV_K3S2 $shape 2>&1 | grep wgrad; done
done
