#!/bin/bash
# round 5, lease 4: is the ring kernel's epilogue (19-22 % of a matrix wave's life) bound by its instructions or by its stores?
# PWS_RING_TIMERS=2 build: every epilogue instruction runs, the 16-byte stores are predicated off
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5d
export CONV_BENCH_STORE16=1 CONV_BENCH_ROTATE=3
for lib in tools/_bin/ring_timers.so tools/_bin/ring_timers_nostore.so; do
  echo "#### $lib"
  export PWS_LIB_PATH=$PWD/$lib
  for shape in "CONVT_K4S2 64 128 128 256 64" "CONV_K3S1 64 128 128 128 128" "CONV_K3S1 64 256 256 64 64"; do
    echo "== $shape fwd"; python tools/ring_timers.py $shape
    echo "== $shape dgrad"; python tools/ring_timers.py $shape dgrad
  done
done > gpurun_out/r5d/timers.txt 2>&1
