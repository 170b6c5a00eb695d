cd ${GRAFT_REPO_ROOT:-.}
export CONV_BENCH_STORE16=1 CONV_BENCH_ROTATE=3
for e in 0 41 42 44 43 45 46 47; do PWS_EXPERIMENT=$e python tools/conv_bench.py CONV_K5S1 64 256 256 32 64 bf16 2>&1 | grep -v amdgpu; done
