#!/bin/bash
# A/B of the bf16 conv kernels on the configs[2] layer shapes (batch 64 = 32 item pairs x 2 forwards), bf16 storage, inputs rotated
# through 3 buffers.  PWS_EXPERIMENT: 20 = first-generation conv_bf16_kernel, 0 = ring kernel (product), 31 / 32 = ring depth variants.
export CONV_BENCH_STORE16=1 CONV_BENCH_ROTATE=3
EXPS="${EXPS:-20 0}"
for e in $EXPS; do
  export PWS_EXPERIMENT=$e
  python tools/conv_bench.py CONVT_K4S2 64 128 128 256 64 bf16          # up_bottom1 forward
  python tools/conv_bench.py CONVT_K4S2 64 128 128 256 64 bf16 dgrad    # up_bottom1 data gradient (k4s2 over dy)
  python tools/conv_bench.py CONV_K3S1 64 128 128 128 128 bf16          # up_bottom1.conv_same
  python tools/conv_bench.py CONV_K3S1 64 256 256 64 64 bf16            # down_bottom1.conv_same
  python tools/conv_bench.py CONV_K3S1 64 32 32 512 512 bf16            # up_bottom3.conv_same
  python tools/conv_bench.py CONV_K3S2 64 256 256 64 64 bf16            # down1
  python tools/conv_bench.py CONV_K3S2 64 128 128 128 128 bf16 dgrad    # down_bottom2 data gradient (sub-pixel classes)
  python tools/conv_bench.py CONVT_K4S2 64 32 32 1024 128 bf16 dgrad    # up_bottom3 data gradient
done 2>&1 | grep -v amdgpu.ids
