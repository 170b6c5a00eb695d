#!/bin/bash
# the round's final evidence: whole GPU suite, smoke, then tools/round_profiles.sh r06
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6m
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r6m/pytest_gpu.txt 2>&1; tail -3 gpurun_out/r6m/pytest_gpu.txt
python __graft_entry__.py --smoke > gpurun_out/r6m/smoke.txt 2>&1; tail -2 gpurun_out/r6m/smoke.txt
bash tools/round_profiles.sh r06 > gpurun_out/r6m/round_profiles.log 2>&1; tail -5 gpurun_out/r6m/round_profiles.log | cut -c1-300
