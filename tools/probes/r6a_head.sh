#!/bin/bash
# round 6, first lease: the field head without scratch (buffer-descriptor loads) -- element check against the first cut, same-box A/B against HEAD's library,
# the whole GPU suite and a default bench line
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6a
python tools/probes/head_v2_check.py > gpurun_out/r6a/head_check.txt 2>&1
for rep in 1 2; do for lib in ab_base.so pwstablenet_amd/libpwstable_hip.so; do for n in 64 8; do for s in bf16 fp32; do
  echo -n "$lib " ; PWS_LIB_PATH=$PWD/$lib python tools/head_bench.py $n $s 2>&1 | grep "field head"; done; done; done; done > gpurun_out/r6a/head_ab.txt
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r6a/pytest_gpu.txt 2>&1
python bench.py > gpurun_out/r6a/bench.json 2> gpurun_out/r6a/bench.err
tail -3 gpurun_out/r6a/pytest_gpu.txt; cat gpurun_out/r6a/head_ab.txt gpurun_out/r6a/head_check.txt
