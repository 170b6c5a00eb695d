#!/bin/bash
# the round's rocprof evidence: kernel-trace stats + PMC passes of the bench command, PMC of the configs[2] step, a default bench run, layer tables
cd ${GRAFT_REPO_ROOT:-.}
bash tools/profile_round.sh r05 > gpurun_out/r5_profile_round.log 2>&1
bash tools/pmc_train.sh r05 > gpurun_out/r5_pmc_train.log 2>&1
python bench.py > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err
python tools/layer_profile.py > gpurun_out/r05_layers_fp32.txt 2>&1
python tools/layer_profile.py --math bf16 > gpurun_out/r05_layers_bf16.txt 2>&1
python tools/train_profile.py --math bf16 --batch 64 --list 400 > gpurun_out/r05_train_launches_bf16_b64.txt 2>&1
