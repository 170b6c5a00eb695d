#!/bin/bash
# rocprofv3 kernel trace of the configs[2] training step (objective kernels included): bash tools/trace_configs2.sh <tag>
TAG=${1:-c2}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/configs2_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 $REPO/tools/configs2_step.py --reps 2 "$@" > $OUT/run.log 2> $OUT/trace.err
cd $REPO
python3 tools/summarize_prof.py $OUT configs2_$TAG $OUT > $OUT/summary.log 2>&1
rm -rf $OUT/trace
cat $OUT/run.log; head -45 $OUT/summary.log
