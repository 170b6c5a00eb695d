#!/bin/bash
# A round's whole rocprof / bench evidence on one box, from the tree as it stands.  usage: bash tools/round_profiles.sh r06
#   0. the CPU guard first (pytest -m "not gpu"): profiles are not cut from a red tree (round-5 review)
#   1. tools/profile_round.sh: kernel-trace stats of the bench command (default / serial fp32 / serial bf16 / training / configs[2]) + PMC passes
#   2. tools/pmc_train.sh: PMC of the configs[2] step per kernel family
#   3. python bench.py (the driver's own command), layer tables, training launch list, kernel timelines, the parity ledger
# then, in the build container:  bash tools/collect_profiles.sh r06 gpurun_out/r06_bench.json   (copies the summaries into profiles/)
TAG=${1:-r06}
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
if ! timeout 1500 python -m pytest tests -x -q -m "not gpu" > gpurun_out/${TAG}_pytest_cpu.txt 2>&1; then
  echo "CPU tests RED: no profiles from this tree"; tail -5 gpurun_out/${TAG}_pytest_cpu.txt; exit 1
fi
tail -1 gpurun_out/${TAG}_pytest_cpu.txt
bash tools/profile_round.sh $TAG > gpurun_out/${TAG}_profile_round.log 2>&1
bash tools/pmc_train.sh $TAG > gpurun_out/${TAG/r0/r}_pmc_train.log 2>&1
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python tools/layer_profile.py > gpurun_out/${TAG}_layers_fp32.txt 2>&1
python tools/layer_profile.py --math bf16 > gpurun_out/${TAG}_layers_bf16.txt 2>&1
python tools/train_profile.py --math bf16 --batch 64 --list 400 > gpurun_out/${TAG}_train_launches_bf16_b64.txt 2>&1
bash tools/timeline.sh gpurun_out/${TAG}_timeline > /dev/null 2>&1
cd ${GRAFT_REPO_ROOT:-.}
python tools/parity_budget.py > gpurun_out/${TAG}_parity_budget.txt 2> gpurun_out/${TAG}_parity_budget.err
tail -3 gpurun_out/${TAG}_bench.json | cut -c1-400
