#!/bin/bash
# the weight gradients on the side queue (product) against one queue for the backward (PWS_EXPERIMENT 16), configs[2] step, alternating
cd ${GRAFT_REPO_ROOT:-.}
for e in 0 16 0 16; do echo "== PWS_EXPERIMENT $e"; PWS_EXPERIMENT=$e python tools/configs2_step.py --reps 4 2>&1 | grep "configs\[2\]"; done > gpurun_out/r5aa_wside.txt
