#!/bin/bash
# PMC counters for the kernels of one forward (serial queue): bash tools/pmc_kernel.sh <tag>
TAG=${1:-x}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof --serial --no-graph"
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY -d $OUT/pmc_sq -o pmc -- $CMD > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU -d $OUT/pmc_sq2 -o pmc -- $CMD > /dev/null 2> $OUT/pmc_sq2.err
rocprofv3 --pmc GRBM_GUI_ACTIVE -d $OUT/pmc_grbm -o pmc -- $CMD > /dev/null 2> $OUT/pmc_grbm.err
cd $REPO
python3 tools/summarize_prof.py $OUT $TAG $OUT > $OUT/summary.log 2>&1
rm -rf $OUT/pmc_sq $OUT/pmc_sq2 $OUT/pmc_grbm   # the raw rocpd databases are tens of MB each
grep -A 22 "^wino_k3s1\|^conv_mfma_kernel<k2s1,convT,tile 1x16x16" $OUT/summary.log | head -80
