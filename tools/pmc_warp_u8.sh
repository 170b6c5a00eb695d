#!/bin/bash
# Issue-slot accounting of the fused 720p uint8 warp (upsample_grid_sample_u8_kernel): vector instructions per pixel and the share of the CUs' cycles in which a
# vector instruction issues, on the smooth and on the random-weight generator's field.  bash tools/pmc_warp_u8.sh  (separate --pmc passes; kernel-trace only)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_warp_u8
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PWS_WARP_U8=1
for F in smooth generator; do
  export PWS_WARP_FIELD=$F
  i=0
  for GROUP in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 180 rocprofv3 --pmc $GROUP -d $OUT/${F}_g$i -o pmc -- python3 $REPO/tools/warp720_probe.py > /dev/null 2> $OUT/${F}_g$i.err
  done
done
cd $REPO
python3 - <<PY
import glob, sqlite3, collections, os
out = "$OUT"
res = collections.defaultdict(dict)
for d in sorted(glob.glob(out + "/*_g*")):
    if not os.path.isdir(d): continue
    f = glob.glob(d + "/**/*.db", recursive=True)
    if not f: print(os.path.basename(d), "no db"); continue
    c = sqlite3.connect(f[0])
    per = collections.defaultdict(lambda: collections.defaultdict(float)); dur = {}
    for name, disp, cn, val, du in c.execute("select kernel_name, dispatch_id, counter_name, value, duration from counters_collection"):
        if "upsample_grid_sample_u8" in name:
            per[disp][cn] += val; dur[disp] = du
    agg = collections.defaultdict(list)
    for disp, cs in per.items():
        for cn, v in cs.items(): agg[cn].append(v)
        agg["_dur_us"].append(dur[disp] / 1e3)
    field = os.path.basename(d).split("_g")[0]
    for cn, v in agg.items(): res[field][cn] = sorted(v)[len(v)//2]
px = 8 * 720 * 1280
for field, r in res.items():
    print("==", field, {k: round(v, 1) for k, v in sorted(r.items())})
    if "SQ_INSTS_VALU" in r:
        # SQ_INSTS_* count wave instructions (one per wave, 64 lanes); a lane owns 4 pixels
        print("   vector instructions per pixel: %.1f (wave instructions %.3g x 64 lanes / %d pixels); scalar per pixel-lane: %.1f; loads %.2f / stores %.2f per pixel" % (
            r["SQ_INSTS_VALU"] * 64 / px, r["SQ_INSTS_VALU"], px, r.get("SQ_INSTS_SALU", 0) * 64 / px, r.get("SQ_INSTS_VMEM_RD", 0) * 64 / px, r.get("SQ_INSTS_VMEM_WR", 0) * 64 / px))
    if "SQ_ACTIVE_INST_VALU" in r and "SQ_BUSY_CYCLES" in r:
        print("   SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES = %.3f ; SQ_ACTIVE_INST_ANY / SQ_BUSY_CYCLES = %.3f" % (r["SQ_ACTIVE_INST_VALU"] / r["SQ_BUSY_CYCLES"], r.get("SQ_ACTIVE_INST_ANY", 0) / r["SQ_BUSY_CYCLES"]))
    if "SQ_WAIT_INST_ANY" in r and "SQ_WAVE_CYCLES" in res[field]:
        pass
PY
rm -rf $OUT/*_g*/
