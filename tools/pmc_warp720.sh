#!/bin/bash
# HBM / L2 traffic of the fused 720p warp on the smooth and on the random-weight generator's field: bash tools/pmc_warp720.sh
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_warp720
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for F in smooth generator; do
  export PWS_WARP_FIELD=$F
  i=0
  for GROUP in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 180 rocprofv3 --pmc $GROUP -d $OUT/${F}_g$i -o pmc -- python3 $REPO/tools/warp720_probe.py > /dev/null 2> $OUT/${F}_g$i.err
  done
done
cd $REPO
python3 - <<PY
import glob, sqlite3, collections, os
out = "$OUT"
for d in sorted(glob.glob(out + "/*_g*")):
    if not os.path.isdir(d): continue
    f = glob.glob(d + "/**/*.db", recursive=True)
    if not f: print(os.path.basename(d), "no db"); continue
    c = sqlite3.connect(f[0])
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='view' or type='table'")]
    per = collections.defaultdict(lambda: collections.defaultdict(float)); dur = {}
    for name, disp, cn, val, du in c.execute("select kernel_name, dispatch_id, counter_name, value, duration from counters_collection"):
        if "upsample_grid_sample" in name:
            per[disp][cn] += val; dur[disp] = du
    agg = collections.defaultdict(list)
    for disp, cs in per.items():
        for cn, v in cs.items(): agg[cn].append(v)
        agg["_dur_us"].append(dur[disp] / 1e3)
    print(os.path.basename(d), {cn: round(sorted(v)[len(v)//2], 1) for cn, v in agg.items()})
PY
rm -rf $OUT/*_g*/
