#!/bin/bash
# round 5, lease 3: the ring kernel's nested-loop loader -- parity (ring tests), then timers and timings of the kinds it was written for
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5c
python -m pytest tests/test_hip_ring.py tests/test_hip_bf16.py -x -q > gpurun_out/r5c/tests.log 2>&1; echo "tests rc $?" > gpurun_out/r5c/rc.txt
export CONV_BENCH_STORE16=1 CONV_BENCH_ROTATE=3
(
export PWS_LIB_PATH=$PWD/tools/_bin/ring_timers.so
for shape in "CONVT_K4S2 64 128 128 256 64" "CONV_K3S2 64 256 256 64 128" "CONV_K3S2 64 128 128 256 256" "CONV_K3S2 64 64 64 512 256" "CONV_K3S1 64 128 128 128 128"; do
  echo "== $shape fwd"; python tools/ring_timers.py $shape
  echo "== $shape dgrad"; python tools/ring_timers.py $shape dgrad
done
) > gpurun_out/r5c/timers.txt 2>&1
(
for e in 0 47 0; do
  export PWS_EXPERIMENT=$e
  for shape in "CONVT_K4S2 64 128 128 256 64" "CONVT_K4S2 64 64 64 512 128" "CONV_K3S2 64 256 256 64 128" "CONV_K3S2 64 128 128 256 256" "CONV_K3S2 64 64 64 512 256" "CONV_K3S1 64 128 128 128 128" "CONV_K3S1 64 256 256 64 64"; do
  python tools/conv_bench.py $shape bf16
  python tools/conv_bench.py $shape bf16 dgrad
  done
done
) 2>&1 | grep -v amdgpu.ids > gpurun_out/r5c/bench.txt
python tools/configs2_step.py > gpurun_out/r5c/configs2.txt 2>&1
python tools/configs2_step.py >> gpurun_out/r5c/configs2.txt 2>&1
