#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5g
python -m pytest tests -m gpu -x -q > gpurun_out/r5g/gpu_tests.log 2>&1; echo "tests rc $?" > gpurun_out/r5g/rc.txt
python bench.py > gpurun_out/r5g/bench.json 2> gpurun_out/r5g/bench.err; echo "bench rc $?" >> gpurun_out/r5g/rc.txt
python tools/layer_profile.py > gpurun_out/r5g/layers_fp32.txt 2>&1
python tools/layer_profile.py --math bf16 > gpurun_out/r5g/layers_bf16.txt 2>&1
