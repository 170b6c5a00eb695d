timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
timeout 900 python bench.py > gpurun_out/bench_now.json 2> gpurun_out/bench_now.err
tail -c 600 gpurun_out/bench_now.err
