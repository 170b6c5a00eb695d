#!/bin/bash
# round 6: conv_ring_kernel's units of 256 / 128 pixels -- parity (ring tests, bf16 tests), per-layer A/B against PWS_EXPERIMENT 185 (no small units), end to end
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6c
timeout 1200 python -m pytest tests/test_hip_ring.py -m gpu -x -q > gpurun_out/r6c/pytest_ring.txt 2>&1
tail -3 gpurun_out/r6c/pytest_ring.txt
export CONV_BENCH_STORE16=1 CONV_BENCH_ROTATE=3
for e in 185 0 181; do
  export PWS_EXPERIMENT=$e
  # training batch 64 (configs[2]): the 16^2 / 8^2 / 4^2 levels
  python tools/conv_bench.py CONV_K3S1 64 16 16 256 256 bf16          # down_bottom5.conv_same
  python tools/conv_bench.py CONV_K3S1 64 16 16 256 256 bf16 dgrad
  python tools/conv_bench.py CONV_K3S2 64 16 16 512 256 bf16          # down_bottom5
  python tools/conv_bench.py CONV_K3S2 64 16 16 512 256 bf16 dgrad
  python tools/conv_bench.py CONVT_K3S1 64 8 8 512 512 bf16           # up_bottom5.conv_same
  python tools/conv_bench.py CONVT_K3S1 64 8 8 512 512 bf16 dgrad
  python tools/conv_bench.py CONVT_K4S2 64 8 8 1024 256 bf16          # up_bottom5
  python tools/conv_bench.py CONVT_K4S2 64 8 8 1024 256 bf16 dgrad
  python tools/conv_bench.py CONVT_K4S2 64 8 8 512 256 bf16           # up5
  python tools/conv_bench.py CONV_K3S1 64 8 8 256 256 bf16            # down_bottom6.conv_same
  python tools/conv_bench.py CONV_K3S2 64 8 8 512 256 bf16            # down_bottom6
  python tools/conv_bench.py CONVT_K4S2 64 4 4 1024 256 bf16          # up_bottom6
  python tools/conv_bench.py CONVT_K3S1 64 4 4 512 512 bf16           # up_bottom6.conv_same
  # inference batch 8 / lockstep 16
  python tools/conv_bench.py CONV_K3S2 8 128 128 64 128 bf16          # down2
  python tools/conv_bench.py CONV_K3S2 8 64 64 128 256 bf16           # down3
  python tools/conv_bench.py CONV_K3S2 16 64 64 256 256 bf16          # down_bottom3 (lockstep)
  python tools/conv_bench.py CONV_K3S1 16 32 32 256 256 bf16          # down_bottom4.conv_same
  python tools/conv_bench.py CONV_K3S2 16 32 32 512 256 bf16          # down_bottom4
  python tools/conv_bench.py CONV_K3S1 16 16 16 256 256 bf16          # down_bottom5.conv_same
  python tools/conv_bench.py CONVT_K3S1 16 16 16 512 512 bf16         # up_bottom4.conv_same
  python tools/conv_bench.py CONVT_K4S2 16 16 16 1024 128 bf16        # up_bottom4
  python tools/conv_bench.py CONVT_K4S2 8 32 32 512 128 bf16          # up4... (stage 1)
  python tools/conv_bench.py CONVT_K4S2 8 16 16 512 256 bf16          # up5
done 2>&1 | grep -v amdgpu.ids > gpurun_out/r6c/conv_ab.txt
unset PWS_EXPERIMENT
python tools/bf16_infer_ab.py 185 0 181 > gpurun_out/r6c/bf16_infer.txt 2>&1
for e in 185 0 185 0; do echo "== PWS_EXPERIMENT $e"; PWS_EXPERIMENT=$e python tools/configs2_step.py 2>&1 | tail -1; done > gpurun_out/r6c/configs2.txt 2>&1
timeout 1200 python -m pytest tests/test_hip_bf16.py tests/test_hip_netg.py tests/test_hip_backward.py tests/test_hip_timed_path.py -m gpu -x -q > gpurun_out/r6c/pytest_rest.txt 2>&1
tail -3 gpurun_out/r6c/pytest_rest.txt
cat gpurun_out/r6c/bf16_infer.txt gpurun_out/r6c/configs2.txt | grep -v amdgpu; 
