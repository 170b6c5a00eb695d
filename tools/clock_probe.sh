#!/bin/bash
# engine clock and power under ONE kind of launch repeated for 8 s (conv_bench.py CONV_BENCH_SECONDS): is the matrix pipe's peak the 2.4 GHz one?
cd ${GRAFT_REPO_ROOT:-.}
export CONV_BENCH_STORE16=1 CONV_BENCH_ROTATE=3 CONV_BENCH_SECONDS=8
rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power (W)"
for e in 0 105 41 42; do
  PWS_EXPERIMENT=$e python tools/conv_bench.py CONV_K3S1 64 128 128 128 128 bf16 2>&1 | grep -v amdgpu.ids
done
PWS_EXPERIMENT=0 python tools/conv_bench.py CONV_K3S1 8 256 256 64 64 2>&1 | grep -v amdgpu.ids
