#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r5ad
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_bf16.py -x -q -m gpu -k "head or field or netg" > $O/tests.log 2>&1; echo "rc $?" > $O/rc.txt
for st in bf16 fp32; do for n in 64 8; do for e in 0 35 0 35; do PWS_EXPERIMENT=$e python tools/head_bench.py $n $st 2>&1 | grep "field head"; done; done; done > $O/head.txt
python tools/fp32_infer_ab.py 0 35 0 35 2>&1 | tail -4 >> $O/head.txt
python tools/bf16_infer_ab.py 0 35 0 35 2>&1 | tail -4 >> $O/head.txt
for e in 0 35 0 35; do PWS_EXPERIMENT=$e python tools/configs2_step.py --reps 4 2>&1 | grep "configs\[2\]" | cut -c1-60; done >> $O/head.txt
