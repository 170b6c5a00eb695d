#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5f
python -m pytest tests/test_hip_bf16.py -x -q -k "one_pass or training_step or off_the_32" > gpurun_out/r5f/tests.log 2>&1; echo "tests rc $?" > gpurun_out/r5f/rc.txt
for e in 120 0 120 0 120 0; do
  echo "== PWS_EXPERIMENT=$e"; PWS_EXPERIMENT=$e python tools/configs2_step.py 2>&1 | grep "configs\[2\]"
done > gpurun_out/r5f/ab.txt 2>&1
