#!/bin/bash
# PMC counters of one conv_bench launch configuration: bash tools/pmc_conv.sh <tag> KIND N H W CIN COUT [wino|bf16]
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmcconv_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $REPO/tools/conv_bench.py $*"
# one counter group per run, each under its own timeout (a profiler abort otherwise hangs until the box limit)
timeout 240 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $CMD > /dev/null 2> $OUT/pmc_fetch.err
timeout 240 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $CMD > /dev/null 2> $OUT/pmc_write.err
timeout 240 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/pmc_sq -o pmc -- $CMD > /dev/null 2> $OUT/pmc_sq.err
timeout 240 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU -d $OUT/pmc_sq2 -o pmc -- $CMD > /dev/null 2> $OUT/pmc_sq2.err
timeout 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $OUT/pmc_tcc -o pmc -- $CMD > /dev/null 2> $OUT/pmc_tcc.err
timeout 240 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TA_TA_BUSY_sum -d $OUT/pmc_tcp -o pmc -- $CMD > /dev/null 2> $OUT/pmc_tcp.err
timeout 240 rocprofv3 --pmc GRBM_GUI_ACTIVE -d $OUT/pmc_grbm -o pmc -- $CMD > /dev/null 2> $OUT/pmc_grbm.err
cd $REPO
python3 tools/summarize_prof.py $OUT $TAG $OUT > $OUT/summary.log 2>&1
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq $OUT/pmc_sq2 $OUT/pmc_grbm $OUT/pmc_tcc $OUT/pmc_tcp
grep -A 44 "conv_ring_kernel\|^conv_bf16_kernel\|^conv_mfma_kernel\|^wino" $OUT/summary.log | head -120
