#!/bin/bash
# bench.py --gpus 2 on one device after the in-flight change + the timed-path tests + the threads tests
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6s
timeout 1400 python -m pytest tests/test_hip_timed_path.py tests/test_hip_threads.py tests/test_hip_netg.py -x -q > gpurun_out/r6s/pytest.txt 2>&1; tail -4 gpurun_out/r6s/pytest.txt
