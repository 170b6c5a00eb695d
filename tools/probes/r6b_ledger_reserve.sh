#!/bin/bash
# round 6: parity ledger (tools/parity_budget.py), the CU reserve for persistent launches beside the other queue (PWS_OPT_EXPERIMENT 170 + k), the review's new tests
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6b
python tools/parity_budget.py > gpurun_out/r6b/parity_budget.txt 2> gpurun_out/r6b/parity_budget.err
python tools/fp32_infer_ab.py 0 171 172 173 174 > gpurun_out/r6b/fp32_reserve.txt 2>&1
python tools/bf16_infer_ab.py 0 171 172 173 174 > gpurun_out/r6b/bf16_reserve.txt 2>&1
timeout 900 python -m pytest tests/test_hip_bf16.py tests/test_hip_backward.py tests/test_hip_timed_path.py -m gpu -x -q -k "wide_windows or overlapped or configs1" > gpurun_out/r6b/pytest.txt 2>&1
tail -3 gpurun_out/r6b/pytest.txt; cat gpurun_out/r6b/parity_budget.txt gpurun_out/r6b/fp32_reserve.txt gpurun_out/r6b/bf16_reserve.txt | grep -v amdgpu.ids
