#!/bin/bash
# queue schedules of the lockstep forward (csrc/netg.cpp forward_lockstep, PWS_EXPERIMENT 150 + v) against the per-stage schedule (15)
cd ${GRAFT_REPO_ROOT:-.}
python tools/fp32_infer_ab.py 15 150 151 152 153 154 2>&1 | tail -6
python tools/bf16_infer_ab.py 15 150 151 152 153 154 2>&1 | tail -6
