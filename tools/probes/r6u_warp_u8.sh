#!/bin/bash
# the rewritten uint8 720p warp: its tests, the A/B switches, the three fields (tools/warp_u8_ab.py), the 720p legs of the bench
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6u
timeout 1200 python -m pytest tests/test_hip_stream.py tests/test_hip_edges.py tests/test_hip_sweep.py tests/test_hip_threads.py -x -q > gpurun_out/r6u/pytest.txt 2>&1; tail -3 gpurun_out/r6u/pytest.txt
python tools/probes/r6v_warp_u8_ablate.py 2>/dev/null | tee gpurun_out/r6u/switches.txt
python tools/warp_u8_ab.py 2>/dev/null | tee gpurun_out/r6u/warp_u8_ab.txt
python bench.py --no-cpu-baseline 2>/dev/null > gpurun_out/r6u/bench.json; python -c "
import json
d=json.loads(open('gpurun_out/r6u/bench.json').read().strip().splitlines()[-1])
print('value',d['value'],'720p',d['value_720p']['value'],'u8',d['value_720p_u8']['value'], json.dumps(d['value_720p_u8']['roofline_warp'])[:600])
print('stream', json.dumps(d.get('value_720p_stream_u8', d.get('streaming', {})))[:400])"
