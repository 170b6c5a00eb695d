#!/bin/bash
# where the GPU idles inside one configs[2] training step (eager launches from the host thread + the autograd thread)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5t
mkdir -p $O
timeout 900 rocprofv3 --kernel-trace -d $O/tr -o trace -- python3 $R/tools/configs2_step.py --reps 3 > $O/run.log 2>&1
python3 $R/tools/timeline_gaps.py $O/tr adam_multi 40 > $O/gaps.txt 2>&1
rm -rf $O/tr
