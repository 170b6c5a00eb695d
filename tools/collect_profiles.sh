#!/bin/bash
# Copies the summaries of tools/profile_round.sh <tag> + tools/pmc_train.sh <tag> + a default bench run out of gpurun_out/ (scratch)
# into profiles/ (tracked), under the names profiles/README.md documents.  usage: bash tools/collect_profiles.sh r03 [bench.json]
set -e
TAG=${1:-r03}
P=gpurun_out/prof_$TAG
T=gpurun_out/pmc_train_$TAG
cp $P/${TAG}_kernel_stats.csv $P/${TAG}_kernel_stats_bf16.csv $P/${TAG}_kernel_stats_train_bf16.csv $P/${TAG}_kernel_stats_configs2_bf16.csv $P/${TAG}_pmc.json profiles/
cp $P/trace_bench.json profiles/${TAG}_bench_serial_under_rocprof.json
cp $P/trace_default_bench.json profiles/${TAG}_bench_default_under_rocprof.json
cp $P/trace_bf16_bench.json profiles/${TAG}_bench_bf16_serial_under_rocprof.json
cp $P/trace_configs2_run.log profiles/${TAG}_configs2_step_under_rocprof.log
cp $T/train_${TAG}_pmc.json profiles/${TAG}_pmc_train_bf16.json
grep -A40 "^kernel  " gpurun_out/${TAG/r0/r}_pmc_train.log > profiles/${TAG}_pmc_train_bf16_table.log || true
[ -n "$2" ] && cp "$2" profiles/${TAG}_bench.json
ls -la profiles/${TAG}_*
