#!/bin/bash
# does the round-3 tree (tools/_bin/r03tree) show the sporadic slow mode of the configs[2] tool as well?
cd ${GRAFT_REPO_ROOT:-.}
for i in $(seq 1 12); do echo -n "r04: "; python tools/configs2_step.py --reps 6 2>&1 | tail -1; echo -n "r03: "; (cd tools/_bin/r03tree && python tools/configs2_step.py --reps 6 2>&1 | tail -1); done
