#!/bin/bash
# Do the other batch's short launches lengthen the persistent kernels?  rocprofv3 --kernel-trace --stats of the fp32 headline step at --in-flight 1
# (one step at a time, ONE queue: --serial) and --in-flight 2: total time per kernel family.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r7b
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for d in 1 2; do
  if [ $d = 1 ]; then extra="--serial"; else extra="--in-flight 2"; fi
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/tr_$d -o trace -- python3 $R/bench.py --no-cpu-baseline --no-extra --no-prof --steps 60 --warmup 10 $extra > $O/run_$d.log 2>&1
  f=$(find $O/tr_$d -name "*.db" | head -1)
  echo "== in-flight $d: $(grep -o '"value": [0-9.]*' $O/run_$d.log | head -1)"
  python3 - "$f" <<'PY'
import sqlite3, sys, collections
tot = collections.defaultdict(lambda: [0, 0.0])
for n, dur in sqlite3.connect(sys.argv[1]).execute("select name, duration from kernels"):
    key = "wino_ring" if "wino_ring" in n else "conv_ringf" if "conv_ringf" in n else "wino5_first" if "wino5_first" in n else "conv_mfma(small)" if "conv_mfma" in n else "conv_skinny" if "skinny" in n else "splitk_reduce" if "splitk" in n else "heads+warp" if ("head" in n or "grid_sample" in n) else "other"
    tot[key][0] += 1; tot[key][1] += dur / 1e6
for k, (c, ms) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("   %-18s %6d launches %9.2f ms" % (k, c, ms))
print("   sum %.2f ms" % sum(v[1] for v in tot.values()))
PY
  rm -rf $O/tr_$d
done
