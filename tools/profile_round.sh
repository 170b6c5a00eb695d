#!/bin/bash
# rocprofv3 evidence for one round: kernel-trace stats of the bench command + PMC passes (each in its own run).
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh r01
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (a) the default bench command (two queues: kernels of the two branches overlap, so their durations are inflated)
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace_default -o trace -- python3 $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $OUT/trace_default_bench.json 2> $OUT/trace_default.err
# (b) the same bench on a single queue, eager launches: clean per-kernel durations (what bench.py's roofline reports)
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --serial --no-graph > $OUT/trace_bench.json 2> $OUT/trace.err
# (c) the bf16-math variant of (b): per-kernel durations of the bf16 matrix-core path
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace_bf16 -o trace -- python3 $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --serial --no-graph --math bf16 > $OUT/trace_bf16_bench.json 2> $OUT/trace_bf16.err
# (d) one configs[2]-shaped training step (batch 32, bf16 math + storage): forward is_training + backward + Adam
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace_train -o trace -- python3 $REPO/tools/train_profile.py --reps 2 --math bf16 --batch 32 > $OUT/trace_train_run.log 2> $OUT/trace_train.err
# (e) BASELINE configs[2] itself as bench.py times it: objective.train_step on 32 item pairs = 64 forwards, objective kernels included
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace_configs2 -o trace -- python3 $REPO/tools/configs2_step.py --reps 2 > $OUT/trace_configs2_run.log 2> $OUT/trace_configs2.err
SHORT="python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-prof --serial --no-graph --gs-batch 256"
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $SHORT > /dev/null 2> $OUT/pmc_fetch.err
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $SHORT > /dev/null 2> $OUT/pmc_write.err
timeout 600 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY -d $OUT/pmc_sq -o pmc -- $SHORT > /dev/null 2> $OUT/pmc_sq.err
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d $OUT/pmc_grbm -o pmc -- $SHORT > /dev/null 2> $OUT/pmc_grbm.err
cd $REPO
python3 tools/summarize_prof.py $OUT $TAG > $OUT/summary.log 2>&1
rm -rf $OUT/trace $OUT/trace_default $OUT/trace_bf16 $OUT/trace_train $OUT/trace_configs2 $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq $OUT/pmc_grbm   # raw rocpd databases: tens of MB each
ls -R $OUT | head -50
tail -40 $OUT/summary.log
