#!/bin/bash
# same-box A/B: tools/_bin/ab_base.so (previous conv_ring.hip) against the tree's library, alternating; ring tests first
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5e
python -m pytest tests/test_hip_ring.py tests/test_hip_bf16.py -x -q > gpurun_out/r5e/tests.log 2>&1; echo "tests rc $?" > gpurun_out/r5e/rc.txt
export CONV_BENCH_STORE16=1 CONV_BENCH_ROTATE=3
for lib in tools/_bin/ab_base.so pwstablenet_amd/libpwstable_hip.so tools/_bin/ab_base.so pwstablenet_amd/libpwstable_hip.so; do
  echo "== $lib"
  export PWS_LIB_PATH=$PWD/$lib
  for shape in ${SHAPES:-"CONVT_K4S2 64 128 128 256 64" "CONVT_K4S2 64 64 64 512 128" "CONV_K3S1 64 128 128 128 128" "CONV_K3S1 64 256 256 64 64" "CONV_K3S2 64 256 256 64 128" "CONV_K3S2 64 128 128 256 256"}; do
    python tools/conv_bench.py $shape bf16
    python tools/conv_bench.py $shape bf16 dgrad
  done 2>&1 | grep -v amdgpu.ids
  python tools/configs2_step.py 2>&1 | grep "configs\[2\]"
  python tools/configs2_step.py 2>&1 | grep "configs\[2\]"
done > gpurun_out/r5e/ab.txt 2>&1
