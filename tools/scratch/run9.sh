timeout 900 python -m pytest tests/test_hip_perceptual.py tests/test_hip_objective.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/bench_now2.json 2> gpurun_out/bench_now2.err; tail -c 300 gpurun_out/bench_now2.err
