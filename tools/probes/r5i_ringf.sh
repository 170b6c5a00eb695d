#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5i
python -m pytest tests/test_hip_ops.py tests/test_hip_netg.py tests/test_hip_edges.py tests/test_hip_timed_path.py -q -x -k "not bench_self" > gpurun_out/r5i/tests.log 2>&1; echo "tests rc $?" > gpurun_out/r5i/rc.txt
for lib in tools/_bin/ab_base.so pwstablenet_amd/libpwstable_hip.so tools/_bin/ab_base.so pwstablenet_amd/libpwstable_hip.so; do
  echo "== $lib"
  PWS_LIB_PATH=$PWD/$lib python tools/fp32_infer_ab.py 0 2>&1 | grep "exp"
  PWS_LIB_PATH=$PWD/$lib python tools/layer_profile.py 2>&1 | grep "conv_ringf\|sum of kernel"
done > gpurun_out/r5i/ab.txt 2>&1
