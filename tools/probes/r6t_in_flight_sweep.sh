#!/bin/bash
# how many batches in flight: the headline step (fp32 and bf16 math) at --in-flight 1..4, same box
cd ${GRAFT_REPO_ROOT:-.}
for m in fp32 bf16; do for d in 1 2 3 4 2 3; do
python bench.py --no-cpu-baseline --no-extra --no-prof --math $m --in-flight $d --steps 120 --warmup 12 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m in-flight $d: value', d['value'], 'ms/step', d['ms_per_step'], '720p', d['value_720p']['value'])"
done; done
