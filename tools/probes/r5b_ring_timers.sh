#!/bin/bash
# round 5, lease 2: (a) where the 2x2-tap kinds of the ring kernel spend their time (in-kernel timers + timing-only ablations),
# (b) which output of the deterministic temporal backward differs beside the ring kernel
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5b
export CONV_BENCH_STORE16=1 CONV_BENCH_ROTATE=3
(
export PWS_LIB_PATH=$PWD/tools/_bin/ring_timers.so
for shape in "CONVT_K4S2 64 128 128 256 64" "CONVT_K4S2 64 64 64 512 128" "CONV_K3S2 64 256 256 64 128" "CONV_K3S2 64 128 128 256 256" "CONV_K3S2 64 64 64 512 256" "CONV_K3S1 64 128 128 128 128"; do
  echo "== $shape fwd"; python tools/ring_timers.py $shape
  echo "== $shape dgrad"; python tools/ring_timers.py $shape dgrad
done
) > gpurun_out/r5b/timers.txt 2>&1
(
for e in 0 41 42 44 43 45 46 47 0; do
  export PWS_EXPERIMENT=$e
  python tools/conv_bench.py CONVT_K4S2 64 128 128 256 64 bf16
  python tools/conv_bench.py CONVT_K4S2 64 128 128 256 64 bf16 dgrad
  python tools/conv_bench.py CONV_K3S2 64 256 256 64 128 bf16 dgrad
  python tools/conv_bench.py CONV_K3S2 64 128 128 256 256 bf16
  python tools/conv_bench.py CONV_K3S2 64 128 128 256 256 bf16 dgrad
done
) 2>&1 | grep -v amdgpu.ids > gpurun_out/r5b/ablations.txt
PROBE_DETAIL=2 PROBE_LAUNCHES=12000 timeout 300 python tools/probes/kernel_victim_probe.py 2.5 conv > gpurun_out/r5b/temporal_detail.log 2>&1
