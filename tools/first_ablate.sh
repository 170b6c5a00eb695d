#!/bin/bash
# Timing-only ablations of conv_first_kernel (PWS_EXPERIMENT = 1100 + mask: 1 no operand reads, 2 no DMA, 4 no stores, 8 no barriers)
for e in 0 1101 1102 1104 1108 1103 1107 1115 25; do
  export PWS_EXPERIMENT=$e
  python tools/conv_bench.py first 2>&1 | tail -1
done
