cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_hip_timed_path.py tests/test_hip_threads.py -m gpu -x -q 2>&1 | tail -3
python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value',d['value'],'one_in_flight',d.get('value_one_in_flight',{}).get('value'),'pruned',d.get('value_dead_layer_pruned',{}).get('value'))
b=d['bf16']['inference']; print('bf16',b['value'],b.get('value_one_in_flight'),b.get('batches_in_flight'))
print(d['config']['launch']); print('720p', d['value_720p']['value'], d['value_720p_u8']['value'], 'configs2', d['bf16']['training_step']['configs2_bf16_batch32']['ms_per_step'])"
python bench.py --no-cpu-baseline --no-extra --in-flight 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('in-flight 1: value', d['value'], d['config']['launch'][:60])"
