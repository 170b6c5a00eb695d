#!/bin/bash
# bf16 conv kernel sweep over the generator's main layer shapes (batch 8): bash tools/bf16_sweep.sh
for sh in "CONV_K3S1 8 256 256 64 64" "CONV_K3S1 8 128 128 128 128" "CONV_K3S1 8 64 64 256 256" "CONV_K3S1 8 32 32 512 512" \
          "CONV_K3S1 8 16 16 512 512" "CONV_K3S2 8 256 256 64 64" "CONV_K3S2 8 128 128 192 128" "CONV_K3S2 8 64 64 384 256" \
          "CONV_K3S2 8 32 32 768 256" "CONV_K3S2 8 8 8 512 256" "CONVT_K4S2 8 128 128 256 64" "CONVT_K4S2 8 64 64 512 64" \
          "CONVT_K4S2 8 32 32 1024 128" "CONVT_K4S2 8 16 16 1024 256" "CONVT_K4S2 8 4 4 1024 256"; do
  timeout 120 python tools/conv_bench.py $sh bf16 2>&1 | grep TFLOP
done
