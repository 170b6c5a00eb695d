mkdir -p gpurun_out/r5a
python -m pytest tests -m gpu -x -q > gpurun_out/r5a/gpu_tests.log 2>&1; echo "tests rc $?" > gpurun_out/r5a/rc.txt
for ag in none conv poison poison128 poisonlds poisonhold; do
  timeout 300 python tools/probes/kernel_victim_probe.py 2.5 $ag > gpurun_out/r5a/victim_$ag.log 2>&1
  echo "$ag rc $?" >> gpurun_out/r5a/rc.txt
done
PROBE_DETAIL=1 PROBE_LAUNCHES=6000 timeout 200 python tools/probes/kernel_victim_probe.py 2.5 poison > gpurun_out/r5a/victim_poison_detail.log 2>&1
python bench.py > gpurun_out/r5a/bench.json 2> gpurun_out/r5a/bench.err; echo "bench rc $?" >> gpurun_out/r5a/rc.txt
