#!/bin/bash
# (a) wino5_first_kernel with s_setprio round the matrix phase (1332) against without; (b) bf16 batch-8 layer table with the ring kernel forced
# onto launches of fewer than 192 units (PWS_EXPERIMENT 21) against the product dispatch
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5p
O=gpurun_out/r5p
for e in 0 1332 0 1332; do PWS_EXPERIMENT=$e timeout 300 python tools/conv_bench.py first 8 256 256 31 64 2>&1 | grep -v amdgpu.ids; done > $O/bench_first.txt
for e in 0 21; do echo "== PWS_EXPERIMENT $e"; PWS_EXPERIMENT=$e python tools/layer_profile.py --math bf16 2>&1 | grep -v amdgpu.ids; done > $O/layers_bf16.txt
