#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6d
bash tools/timeline.sh gpurun_out/r6d
cd ${GRAFT_REPO_ROOT:-.}
python tools/layer_profile.py --math bf16 > gpurun_out/r6d/layers_bf16.txt 2>&1
python tools/train_profile.py --math bf16 --batch 64 --list 400 > gpurun_out/r6d/train_launches_bf16_b64.txt 2>&1
python tools/bf16_infer_ab.py 0 181 0 181 > gpurun_out/r6d/bf16_infer.txt 2>&1
for e in 181 0 181 0; do echo "== PWS_EXPERIMENT $e"; PWS_EXPERIMENT=$e python tools/configs2_step.py 2>&1 | tail -1; done > gpurun_out/r6d/configs2.txt 2>&1
