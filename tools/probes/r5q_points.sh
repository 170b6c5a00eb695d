#!/bin/bash
# wino5_first_kernel on the points {0, 1, -1, 2, -1/2, inf}: parity, the whole-network bound at batch 8 (with the direct first layer beside it: PWS_EXPERIMENT 26), time
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5q
O=gpurun_out/r5q
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "first_layer or conv_first" > $O/tests_first.log 2>&1; echo "rc first $?" >> $O/rc.txt
timeout 900 python -m pytest tests/test_hip_timed_path.py -x -q -m gpu -k "configs1_graph" -s > $O/tests_timed.log 2>&1; echo "rc timed $?" >> $O/rc.txt
timeout 900 python tools/probes/first_layer_margin.py 2>&1 | grep -v amdgpu.ids > $O/margin.txt
for e in 0 26 0; do PWS_EXPERIMENT=$e timeout 300 python tools/conv_bench.py first 8 256 256 31 64 2>&1 | grep -v amdgpu.ids; done > $O/bench_first.txt
timeout 2400 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "rc all $?" >> $O/rc.txt
python bench.py > $O/bench.json 2> $O/bench.err
