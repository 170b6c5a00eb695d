#!/bin/bash
# whole GPU suite + the driver's bench command, from the tree as it stands (no profiles)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6z
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r6z/pytest_gpu.txt 2>&1; tail -3 gpurun_out/r6z/pytest_gpu.txt
python __graft_entry__.py --smoke > gpurun_out/r6z/smoke.txt 2>&1; tail -1 gpurun_out/r6z/smoke.txt
python bench.py > gpurun_out/r6z/bench.json 2> gpurun_out/r6z/bench.err; python -c "
import json
d=json.loads(open('gpurun_out/r6z/bench.json').read().strip().splitlines()[-1])
print('value',d['value'],d['value_one_in_flight']['value'],d['value_dead_layer_pruned']['value'],'720p',d['value_720p']['value'],'u8',d['value_720p_u8']['value'],d['value_720p_u8']['roofline_warp']['frac'])
b=d['bf16']; print('bf16', b['inference']['value'], b['inference'].get('value_one_in_flight'), b['training_step']['configs2_bf16_batch32']['ms_per_step'])
s=d['value_720p_stream_u8']; print('stream', s.get('value'), s.get('value_one_in_flight'), s.get('chunks_in_flight'), s.get('bf16_generator'), s.get('error'))"
