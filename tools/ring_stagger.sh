#!/bin/bash
# Start-time stagger of the ring kernel's workgroups (PWS_EXPERIMENT 110 + k: step of k x 0.5 us between the 4 phases)
export CONV_BENCH_STORE16=1 CONV_BENCH_ROTATE=3
for e in ${EXPS:-0 112 114 116 120 0}; do
  export PWS_EXPERIMENT=$e
  python tools/conv_bench.py CONV_K3S1 64 256 256 64 64 bf16
  python tools/conv_bench.py CONV_K3S1 64 128 128 128 128 bf16
  python tools/conv_bench.py CONVT_K4S2 64 128 128 256 64 bf16
  python tools/conv_bench.py CONVT_K4S2 64 128 128 256 64 bf16 dgrad
done 2>&1 | grep -v amdgpu.ids
