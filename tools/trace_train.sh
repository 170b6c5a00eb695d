#!/bin/bash
# rocprofv3 kernel trace of one training step profile run: bash tools/trace_train.sh <tag> [--math bf16]
TAG=${1:-t}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/train_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 $REPO/tools/train_profile.py --reps 2 "$@" > $OUT/run.log 2> $OUT/trace.err
cd $REPO
python3 tools/summarize_prof.py $OUT train_$TAG $OUT > $OUT/summary.log 2>&1
rm -rf $OUT/trace
head -40 $OUT/summary.log
