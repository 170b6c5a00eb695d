# the 64-channel 3x3 layers' weight gradient: wgrad_bf16_kernel (product) against wgrad_ring_kernel forced (PWS_EXPERIMENT 81), after round 5's loader rewrite
cd ${GRAFT_REPO_ROOT:-.}
for e in 0 81 0 81; do
  PWS_EXPERIMENT=$e python tools/wgrad_bench.py CONV_K3S1 64 256 256 64 64 2>&1 | grep -v amdgpu     # down_bottom1.conv_same
  PWS_EXPERIMENT=$e python tools/wgrad_bench.py CONV_K3S1 64 128 128 64 64 2>&1 | grep -v amdgpu     # down_bottom2.conv_same
  PWS_EXPERIMENT=$e python tools/wgrad_bench.py CONVT_K3S1 64 128 128 128 128 2>&1 | grep -v amdgpu  # (>= 128 channels: ring already)
done
