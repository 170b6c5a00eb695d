timeout 1200 python -m pytest tests/test_hip_ops.py tests/test_hip_bf16.py tests/test_hip_backward.py tests/test_hip_netg.py tests/test_hip_edges.py -x -q -m gpu 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
for e in 9 0; do
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/pmc_ct4_$e
PWS_EXPERIMENT=$e CONV_BENCH_ROTATE=3 rocprofv3 --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/pmc_ct4_$e/pmc_fetch -o pmc -- python3 $GRAFT_REPO_ROOT/tools/conv_bench.py CONVT_K4S2 8 128 128 128 64 > /dev/null 2>&1
PWS_EXPERIMENT=$e CONV_BENCH_ROTATE=3 rocprofv3 --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/pmc_ct4_$e/pmc_write -o pmc -- python3 $GRAFT_REPO_ROOT/tools/conv_bench.py CONVT_K4S2 8 128 128 128 64 > /dev/null 2>&1
( cd $GRAFT_REPO_ROOT && python3 tools/summarize_prof.py gpurun_out/pmc_ct4_$e ct4_$e > /dev/null 2>&1; python3 - <<PY
import json
d = json.load(open("gpurun_out/pmc_ct4_$e/ct4_${e}_pmc.json"))
for k, v in d.items():
    if "conv_mfma" in k: print("exp=$e", k[:80], {c: round(x["mean_per_dispatch"]) for c, x in v.items() if isinstance(x, dict)})
PY
)
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_ct4_$e
done
