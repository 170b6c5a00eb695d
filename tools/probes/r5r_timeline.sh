#!/bin/bash
# kernel timeline of one graph-replayed forward (two queues), fp32 and bf16
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5r
mkdir -p $O
for m in fp32 bf16; do
  timeout 600 rocprofv3 --kernel-trace -d $O/tr_$m -o trace -- python3 $R/tools/probes/infer_loop.py --math $m --reps 12 > $O/run_$m.log 2>&1
  if [ $m = fp32 ]; then mk=wino5_first; else mk=nchw_to_nhwc; fi
  python3 $R/tools/timeline_prof.py $O/tr_$m $mk > $O/timeline_$m.txt 2>&1
  rm -rf $O/tr_$m
done
