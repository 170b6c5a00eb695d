#!/bin/bash
# Ring weight-gradient kernel (PWS_EXPERIMENT 0; 82 = transposed kind with one class per workgroup) vs wgrad_bf16_kernel (80) on the
# layers that dominate the configs[2] step (batch 64)
for e in ${WGR_EXPS:-0 80}; do
  export PWS_EXPERIMENT=$e
  python tools/wgrad_bench.py CONV_K3S1 64 256 256 64 64
  python tools/wgrad_bench.py CONV_K3S1 64 128 128 128 128
  python tools/wgrad_bench.py CONV_K3S1 64 64 64 256 256
  python tools/wgrad_bench.py CONV_K3S1 64 32 32 512 512
  python tools/wgrad_bench.py CONVT_K4S2 64 128 128 256 64
  python tools/wgrad_bench.py CONVT_K4S2 64 64 64 512 64
  python tools/wgrad_bench.py CONVT_K4S2 64 32 32 1024 128
  python tools/wgrad_bench.py CONVT_K4S2 64 16 16 1024 256
  python tools/wgrad_bench.py CONV_K5S1 64 256 256 32 64
done 2>&1 | grep -v amdgpu.ids
