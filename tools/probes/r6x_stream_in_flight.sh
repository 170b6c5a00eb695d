#!/bin/bash
# run_video(in_flight=2): its test, then the bench line (the streaming leg reports both)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r6x
timeout 900 python -m pytest tests/test_hip_stream.py tests/test_hip_timed_path.py -x -q > gpurun_out/r6x/pytest.txt 2>&1; tail -3 gpurun_out/r6x/pytest.txt
python bench.py > gpurun_out/r6x/bench.json 2> gpurun_out/r6x/bench.err; python -c "
import json
d=json.loads(open('gpurun_out/r6x/bench.json').read().strip().splitlines()[-1])
print('value',d['value'],d['value_one_in_flight']['value'],'720p',d['value_720p']['value'],'u8',d['value_720p_u8']['value'])
s=d['value_720p_stream_u8']; print('stream', s.get('value'), s.get('value_one_in_flight'), s.get('chunks_in_flight'), s.get('bf16_generator'), s.get('error'))"
