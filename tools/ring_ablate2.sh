#!/bin/bash
# Timing-only ablations of the ring kernel, second set: 48 = the epilogue's stores all land on the same few lines (its instructions
# run, its HBM write traffic does not), 45 = matrix phase only (no DMA fetch, no epilogue), 44 = no epilogue.
export CONV_BENCH_STORE16=1 CONV_BENCH_ROTATE=3
for e in 0 48 44 45 0 48; do
  export PWS_EXPERIMENT=$e
  python tools/conv_bench.py CONVT_K4S2 64 128 128 256 64 bf16
  python tools/conv_bench.py CONV_K3S1 64 128 128 128 128 bf16
  python tools/conv_bench.py CONV_K3S1 64 256 256 64 64 bf16
done 2>&1 | grep -v amdgpu.ids
