#!/bin/bash
# PMC counters (one group per run) for the kernels of the configs[2] training step: bash tools/pmc_train.sh <tag> [configs2_step.py args]
TAG=${1:-t}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_train_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $REPO/tools/configs2_step.py --reps 1 $@"
timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_BF16 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d $OUT/pmc_sq -o pmc -- $CMD > /dev/null 2> $OUT/pmc_sq.err
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_ANY -d $OUT/pmc_sq2 -o pmc -- $CMD > /dev/null 2> $OUT/pmc_sq2.err
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE -d $OUT/pmc_grbm -o pmc -- $CMD > /dev/null 2> $OUT/pmc_grbm.err
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $CMD > /dev/null 2> $OUT/pmc_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $CMD > /dev/null 2> $OUT/pmc_write.err
cd $REPO
python3 tools/summarize_prof.py $OUT train_$TAG $OUT > $OUT/summary.log 2>&1
rm -rf $OUT/pmc_sq $OUT/pmc_sq2 $OUT/pmc_grbm $OUT/pmc_fetch $OUT/pmc_write
python3 - <<PY
import json
d = json.load(open("$OUT/train_${TAG}_pmc.json"))
rows = []
for k, v in d.items():
    if "GRBM_GUI_ACTIVE" not in v or "SQ_WAVE_CYCLES" not in v: continue
    g = v["GRBM_GUI_ACTIVE"]["mean_per_dispatch"]; n = v["GRBM_GUI_ACTIVE"]["dispatches"]
    m = lambda c: v.get(c, {}).get("mean_per_dispatch", 0.0)
    wc = max(m("SQ_WAVE_CYCLES"), 1.0)
    rows.append((g * n, k, n, g, m("SQ_VALU_MFMA_BUSY_CYCLES") / 1024 / max(g / 8, 1) , m("SQ_WAIT_ANY") / wc, m("SQ_WAIT_INST_ANY") / wc, m("SQ_ACTIVE_INST_ANY") / wc,
                 m("SQ_LDS_BANK_CONFLICT") / max(m("SQ_LDS_IDX_ACTIVE"), 1), (2 * m("FETCH_SIZE") + m("WRITE_SIZE")) * 1024 / 1e6, m("_duration_ns_under_pmc") / 1e3))
print("%-70s %5s %9s %6s %6s %6s %6s %6s %8s %8s" % ("kernel", "n", "cyc/disp", "mfma%", "wait%", "stall%", "act%", "bank%", "HBM MB", "us(pmc)"))
for r in sorted(rows, reverse=True)[:18]:
    print("%-70s %5d %9.0f %6.1f %6.1f %6.1f %6.1f %6.1f %8.1f %8.1f" % (r[1][:70], r[2], r[3], 100 * r[4], 100 * r[5], 100 * r[6], 100 * r[7], 100 * r[8], r[9], r[10]))
PY
