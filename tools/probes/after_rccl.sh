#!/bin/bash
# Does a process that used RCCL leave the GPU slow for the next process?  (round 4: configs[2] step 126 ms instead of 30 right after
# `bench.py --force-collectives`.)  usage on the GPU box: bash tools/probes/after_rccl.sh
cd ${GRAFT_REPO_ROOT:-.}
step() { python tools/configs2_step.py --reps 3 2>&1 | tail -1; }
clk() { rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -3; rocm-smi --showperflevel 2>/dev/null | grep -i perf | head -2; }
echo "== fresh"; step; clk
echo "== after bench (no RCCL)"; python bench.py --no-cpu-baseline --no-extra --steps 5 > /dev/null 2>&1; step
echo "== after a bare one-rank RCCL all_reduce"; python - <<'PY'
import os, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29544")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0), rank=0, world_size=1)
t = torch.ones(1 << 20, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
dist.destroy_process_group()
PY
step; clk
echo "== after bench --force-collectives"; python bench.py --force-collectives --no-cpu-baseline --no-extra --steps 5 > /dev/null 2>&1; step; clk
echo "== 20 s later"; sleep 20; step
ps aux | grep -c python
