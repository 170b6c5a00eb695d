#!/bin/bash
# run the configs[2] tool under rocprofv3 until a slow-mode process shows up; keep the kernel stats of one slow and one normal run
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT}
OUT=$R/gpurun_out/r4slow; mkdir -p $OUT
for i in 1 2 3 4 5 6 7 8 9 10; do
  rm -rf /tmp/cs$i
  rocprofv3 --kernel-trace --stats -d /tmp/cs$i -o t --output-format csv -- python3 $R/tools/configs2_step.py --reps 6 > /tmp/cs$i.log 2>&1
  ms=$(grep "configs\[2\] step" /tmp/cs$i.log | sed 's/.*: \([0-9.]*\) ms.*/\1/')
  echo "run $i: $ms ms"
  f=$(find /tmp/cs$i -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then
    if awk "BEGIN{exit !($ms > 40)}"; then cp $f $OUT/slow_kernel_stats.csv; else cp $f $OUT/normal_kernel_stats.csv; fi
  fi
done
ls $OUT
