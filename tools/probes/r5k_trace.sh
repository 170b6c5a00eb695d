#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r5k
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace_configs2 -o trace -- python3 $REPO/tools/configs2_step.py --reps 3 > $OUT/trace_configs2_run.log 2> $OUT/trace_configs2.err
cd $REPO
python3 tools/summarize_prof.py $OUT r05pre > $OUT/summary.log 2>&1
rm -rf $OUT/trace_configs2
