#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for i in $(seq 1 16); do echo -n "own: "; python tools/configs2_step.py --reps 6 2>&1 | tail -1; echo -n "no-own: "; PWS_NO_OWN=1 python tools/configs2_step.py --reps 6 2>&1 | tail -1; done
