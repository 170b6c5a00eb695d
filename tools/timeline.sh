#!/bin/bash
# Kernel timeline of ONE graph-replayed batch-8 forward (two queues), fp32 and bf16: rocprofv3 --kernel-trace of tools/infer_loop.py through
# tools/timeline_prof.py.   usage: bash tools/timeline.sh <out dir under the repo>   (writes timeline_fp32.txt / timeline_bf16.txt there)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/${1:-gpurun_out/timeline}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in fp32 bf16; do
  timeout 600 rocprofv3 --kernel-trace -d $O/tr_$m -o trace -- python3 $R/tools/infer_loop.py --math $m --reps 12 > $O/run_$m.log 2>&1
  if [ $m = fp32 ]; then mk=wino5_first; else mk=nchw_to_nhwc; fi
  python3 $R/tools/timeline_prof.py $O/tr_$m $mk > $O/timeline_$m.txt 2>&1
  rm -rf $O/tr_$m
done
