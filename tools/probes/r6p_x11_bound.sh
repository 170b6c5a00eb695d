# What one summed data-gradient launch for x11 / x12 could save at most: the second writer's read-modify-write against a plain write (same launch, accumulate 0 / 1),
# batch 64, bf16 storage; and the forward pair's second read of the tensor = one extra pass at the copy rate.
cd ${GRAFT_REPO_ROOT:-.}
export CONV_BENCH_STORE16=1 CONV_BENCH_ROTATE=3
for acc in 0 1 0 1; do
  export CONV_BENCH_ACC=$acc
  echo "== accumulate $acc"
  python tools/conv_bench.py CONV_K3S2 64 256 256 64 64 bf16 dgrad 2>&1 | grep -v amdgpu    # down1 / down_bottom1: dx11
  python tools/conv_bench.py CONV_K3S1 64 256 256 64 64 bf16 dgrad 2>&1 | grep -v amdgpu    # down_bottom1.conv_same: dx11
  python tools/conv_bench.py CONV_K3S2 64 128 128 64 128 bf16 dgrad 2>&1 | grep -v amdgpu   # down2: dx12
  python tools/conv_bench.py CONV_K3S1 64 128 128 64 64 bf16 dgrad 2>&1 | grep -v amdgpu    # down_bottom2.conv_same: dx12
done
