#!/bin/bash
# PMC passes (one counter group per run) over tools/gs_probe.py for a list of variants: bash tools/pmc_gs.sh <case> <variants...>
CASE=${1:-gs256}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_gs_$CASE
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PWS_GS_CASE=$CASE
for V in "$@"; do
  export PWS_GS_VARIANT=$V
  i=0
  for GROUP in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
      "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum" \
      "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM" \
      "GRBM_GUI_ACTIVE GRBM_TA_BUSY" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TAGRAM0_REQ_sum"; do
    i=$((i+1))
    timeout 120 rocprofv3 --pmc $GROUP -d $OUT/v${V}_g$i -o pmc -- python3 $REPO/tools/gs_probe.py > /dev/null 2> $OUT/v${V}_g$i.err
  done
done
cd $REPO
python3 - <<PY
import glob, sqlite3, collections, os
out = "$OUT"
for d in sorted(glob.glob(out + "/v*_g*")):
    if not os.path.isdir(d): continue
    f = glob.glob(d + "/**/*.db", recursive=True)
    if not f: print(os.path.basename(d), "no db"); continue
    c = sqlite3.connect(f[0])
    per = collections.defaultdict(lambda: collections.defaultdict(float)); dur = {}
    for name, disp, cn, val, du in c.execute("select kernel_name, dispatch_id, counter_name, value, duration from counters_collection"):
        if "grid_sample" in name or "warp_tile" in name:
            per[(name.split("(")[0][-60:], disp)][cn] += val; dur[(name.split("(")[0][-60:], disp)] = du
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for (k, disp), cs in per.items():
        for cn, v in cs.items(): agg[k][cn].append(v)
        agg[k]["_dur_ns"].append(dur[(k, disp)])
    for k, cs in agg.items():
        print(os.path.basename(d), k, {cn: round(sum(v)/len(v), 1) for cn, v in cs.items()})
PY
rm -rf $OUT/v*_g*/
