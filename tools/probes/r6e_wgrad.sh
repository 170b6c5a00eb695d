#!/bin/bash
# round 6: wgrad_bf16_kernel -- both operand pairs of a shared layer in one launch (188 = off), at least `mint` tiles per workgroup (190 + k)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6e
timeout 900 python -m pytest tests/test_hip_bf16.py -m gpu -x -q -k "used_twice or weight_gradient or training_step or one_pass" > gpurun_out/r6e/pytest.txt 2>&1
tail -3 gpurun_out/r6e/pytest.txt
for e in 190 192 194 196 198; do
  export PWS_EXPERIMENT=$e
  python tools/wgrad_bench.py CONV_K3S1 64 8 8 256 256      # down_bottom6.conv_same
  python tools/wgrad_bench.py CONV_K3S2 64 8 8 512 256      # down_bottom6
  python tools/wgrad_bench.py CONVT_K3S1 64 8 8 512 512     # up_bottom5.conv_same
  python tools/wgrad_bench.py CONVT_K4S2 64 8 8 1024 256    # up_bottom5
  python tools/wgrad_bench.py CONV_K3S2 64 16 16 512 256    # down_bottom5
  python tools/wgrad_bench.py CONV_K3S2 64 16 16 256 256    # down5
  python tools/wgrad_bench.py CONVT_K3S1 64 4 4 512 512     # up_bottom6.conv_same
  python tools/wgrad_bench.py CONVT_K4S2 64 4 4 1024 256    # up_bottom6
  python tools/wgrad_bench.py CONV_K3S1 64 4 4 256 256      # down_bottom7.conv_same
  python tools/wgrad_bench.py CONVT_K4S2 64 2 2 512 256     # up_bottom7
done 2>&1 | grep -v amdgpu.ids > gpurun_out/r6e/wgrad_mint.txt
unset PWS_EXPERIMENT
for e in 188 0 190 194 198 188 0; do echo "== PWS_EXPERIMENT $e"; PWS_EXPERIMENT=$e python tools/configs2_step.py 2>&1 | tail -1; done > gpurun_out/r6e/configs2.txt 2>&1
cat gpurun_out/r6e/wgrad_mint.txt; cut -c1-80 gpurun_out/r6e/configs2.txt
