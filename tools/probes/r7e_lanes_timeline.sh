#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r7e
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $O/tr -o trace -- python3 $R/bench.py --no-cpu-baseline --no-extra --no-prof --steps 60 --warmup 10 --in-flight 2 --gs-batch 8 > $O/run.log 2>&1
python3 $R/tools/probes/r7e_lanes_timeline.py $O/tr > $O/timeline.txt 2>&1
rm -rf $O/tr
tail -3 $O/timeline.txt
