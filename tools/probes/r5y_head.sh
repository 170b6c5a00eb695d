#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for n in 64 8; do for s in bf16 fp32; do for e in 0 92 90; do PWS_EXPERIMENT=$e python tools/head_bench.py $n $s 2>&1 | grep "field head"; done; done; done > gpurun_out/r5y_head.txt
