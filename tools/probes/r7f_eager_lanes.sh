#!/bin/bash
# is the phase lock of the two lanes a property of graph launches?  eager launches at --in-flight 1 / 2, and the eager timeline
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r7f
mkdir -p $O
cd $R
for d in 1 2 2; do python bench.py --no-graph --no-cpu-baseline --no-extra --no-prof --in-flight $d --steps 120 --warmup 12 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('eager in-flight $d: value', d['value'], 'ms/step', d['ms_per_step'])"; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $O/tr -o trace -- python3 $R/bench.py --no-graph --no-cpu-baseline --no-extra --no-prof --steps 60 --warmup 10 --in-flight 2 --gs-batch 8 > $O/run.log 2>&1
python3 $R/tools/probes/r7e_lanes_timeline.py $O/tr > $O/timeline.txt 2>&1
rm -rf $O/tr
tail -2 $O/timeline.txt
