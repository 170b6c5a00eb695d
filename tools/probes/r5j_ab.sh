#!/bin/bash
# same-box A/B: tools/_bin/ab_base.so (HEAD build) against the tree's library: fp32 ring loader + wgrad prep/issue split
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5j
python -m pytest tests/test_hip_bf16.py tests/test_hip_backward.py -q -k "weight_gradient or used_twice or deterministic" > gpurun_out/r5j/tests.log 2>&1; echo "tests rc $?" > gpurun_out/r5j/rc.txt
for lib in tools/_bin/ab_base.so pwstablenet_amd/libpwstable_hip.so tools/_bin/ab_base.so pwstablenet_amd/libpwstable_hip.so; do
  echo "== $lib"
  export PWS_LIB_PATH=$PWD/$lib
  for shape in "CONVT_K4S2 64 128 128 256 64" "CONVT_K4S2 64 32 32 1024 256" "CONV_K3S1 64 128 128 128 128" "CONV_K3S2 64 256 256 64 128" "CONV_K3S2 64 128 128 256 256"; do
    python tools/wgrad_bench.py $shape
  done 2>&1 | grep -v amdgpu.ids
  python tools/configs2_step.py 2>&1 | grep "configs\[2\]"
  python tools/configs2_step.py 2>&1 | grep "configs\[2\]"
done > gpurun_out/r5j/ab.txt 2>&1
