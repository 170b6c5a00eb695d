#!/bin/bash
# round 6: the bf16 first layer on the ring kernel with resident weights (RM_K5) -- parity, A/B against conv_bf16_k5_kernel (PWS_EXPERIMENT 189)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6g
timeout 600 python -m pytest tests/test_hip_ring.py -m gpu -x -q -k "K5S1" > gpurun_out/r6g/pytest_k5.txt 2>&1; tail -3 gpurun_out/r6g/pytest_k5.txt
timeout 600 python -m pytest tests/test_hip_bf16.py -m gpu -x -q -k "used_twice or first_layer" > gpurun_out/r6g/pytest_b.txt 2>&1; tail -3 gpurun_out/r6g/pytest_b.txt
export CONV_BENCH_STORE16=1 CONV_BENCH_ROTATE=3
for e in 189 0 189 0; do
  PWS_EXPERIMENT=$e python tools/conv_bench.py CONV_K5S1 64 256 256 32 64 bf16
  PWS_EXPERIMENT=$e python tools/conv_bench.py CONV_K5S1 8 256 256 32 64 bf16
done 2>&1 | grep -v amdgpu > gpurun_out/r6g/k5_ab.txt; cat gpurun_out/r6g/k5_ab.txt
python tools/bf16_infer_ab.py 189 0 189 0 2>&1 | grep exp > gpurun_out/r6g/bf16_infer.txt; cat gpurun_out/r6g/bf16_infer.txt
for e in 189 0 189 0; do echo "== $e"; PWS_EXPERIMENT=$e python tools/configs2_step.py 2>&1 | tail -1 | cut -c1-80; done > gpurun_out/r6g/configs2.txt; cat gpurun_out/r6g/configs2.txt
timeout 900 python -m pytest tests/test_hip_bf16.py tests/test_hip_netg.py tests/test_hip_timed_path.py -m gpu -x -q > gpurun_out/r6g/pytest_rest.txt 2>&1; tail -3 gpurun_out/r6g/pytest_rest.txt
