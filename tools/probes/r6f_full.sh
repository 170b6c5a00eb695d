#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6f
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r6f/pytest_gpu.txt 2>&1
tail -3 gpurun_out/r6f/pytest_gpu.txt
python bench.py > gpurun_out/r6f/bench.json 2> gpurun_out/r6f/bench.err
python tools/bf16_infer_ab.py 0 0 > gpurun_out/r6f/bf16_infer.txt 2>&1; grep exp gpurun_out/r6f/bf16_infer.txt
for e in 0 0; do PWS_EXPERIMENT=$e python tools/configs2_step.py 2>&1 | tail -1; done | cut -c1-80
