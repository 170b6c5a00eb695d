#!/bin/bash
# first contact of the F(2x2,5x5) first-layer kernel: parity tests, per-launch time against conv_first_kernel (same box), ablations, the generator's tests, the bench
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5n
O=gpurun_out/r5n
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "first_layer or conv_first" > $O/tests_first.log 2>&1; echo "rc first $?" >> $O/rc.txt
for e in 0 26 0 26; do PWS_EXPERIMENT=$e timeout 300 python tools/conv_bench.py first 8 256 256 31 64 2>&1 | grep -v amdgpu.ids; done > $O/bench_first.txt
for e in 1301 1302 1304 1308 1303 1306 1307; do PWS_EXPERIMENT=$e timeout 300 python tools/conv_bench.py first 8 256 256 31 64 2>&1 | grep -v amdgpu.ids; done > $O/ablate.txt
timeout 1500 python -m pytest tests/test_hip_netg.py -x -q -m gpu > $O/tests_netg.log 2>&1; echo "rc netg $?" >> $O/rc.txt
python tools/layer_profile.py 2>&1 | grep -v amdgpu.ids | head -8 > $O/layers_fp32.txt
python bench.py --no-extra > $O/bench.json 2> $O/bench.err
