#!/bin/bash
# wino5_first_kernel second cut (opposite orders on the two waves of a SIMD) against the first cut (PWS_EXPERIMENT 1316) and conv_first_kernel (26), same box
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5o
O=gpurun_out/r5o
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "first_layer or conv_first" > $O/tests_first.log 2>&1; echo "rc first $?" >> $O/rc.txt
for e in 0 1316 26 0 1316; do PWS_EXPERIMENT=$e timeout 300 python tools/conv_bench.py first 8 256 256 31 64 2>&1 | grep -v amdgpu.ids; done > $O/bench_first.txt
for e in 1301 1302 1304 1308 1303 1306 1307 1315; do PWS_EXPERIMENT=$e timeout 300 python tools/conv_bench.py first 8 256 256 31 64 2>&1 | grep -v amdgpu.ids; done > $O/ablate.txt
python bench.py --no-extra > $O/bench.json 2> $O/bench.err
