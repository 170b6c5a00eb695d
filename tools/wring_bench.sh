#!/bin/bash
# Winograd ring kernel vs the first-generation Winograd kernel on the generator's conv_same layers (batch 8), and the ring
# kernel's timing-only ablations (51 = DMA fetches nothing, 52 = no matrix instructions, 54 = no epilogue stores).
export CONV_BENCH_ROTATE=3
for e in ${WRING_EXPS:-0 50 51 52 54}; do
  export PWS_EXPERIMENT=$e
  python tools/conv_bench.py CONV_K3S1 8 128 128 128 128 wino
  python tools/conv_bench.py CONV_K3S1 8 64 64 256 256 wino
  python tools/conv_bench.py CONV_K3S1 8 32 32 512 512 wino
done 2>&1 | grep -v amdgpu.ids
