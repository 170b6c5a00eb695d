#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for cfg in "40 5" "40 30" "200 30" "40 5" "100 50"; do set -- $cfg; python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('steps $1 warmup $2: %.1f f/s (%.3f ms per batch)' % (d['value'], d['ms_per_step']))"; done > gpurun_out/r5ac_warm.txt 2>&1
