#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5l
python -m pytest tests/test_hip_ring.py tests/test_hip_bf16.py tests/test_hip_backward.py -q > gpurun_out/r5l/tests.log 2>&1; echo "tests rc $?" > gpurun_out/r5l/rc.txt
for i in 1 2 3; do python tools/configs2_step.py 2>&1 | grep "configs\[2\]"; done > gpurun_out/r5l/configs2.txt
