#!/bin/bash
# Timing-only ablations of the ring kernel (results are meaningless): 41 = DMA pieces fetch nothing, 42 = no matrix phase,
# 44 = no epilogue, and their sums (43, 45, 46, 47).
export CONV_BENCH_STORE16=1 CONV_BENCH_ROTATE=3
for e in 0 41 42 44 43 46 47; do
  export PWS_EXPERIMENT=$e
  python tools/conv_bench.py CONVT_K4S2 64 128 128 256 64 bf16
  python tools/conv_bench.py CONV_K3S1 64 128 128 128 128 bf16
  python tools/conv_bench.py CONV_K3S1 64 256 256 64 64 bf16
done 2>&1 | grep -v amdgpu.ids
