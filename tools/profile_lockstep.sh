#!/bin/bash
# rocprofv3 kernel stats + HBM counters for the lockstep kernel k_step (run on the GPU box through gpurun).
#   bash tools/profile_lockstep.sh [B A T N]  ->  gpurun_out/prof_lockstep/
set -u
REPO=$(pwd)
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/prof_lockstep
rm -rf $OUT; mkdir -p $OUT
ARGS="${1:-65536} ${2:-20} ${3:-50} ${4:-40}"
CMD="python3 $REPO/tools/lockstep_probe.py $ARGS"
echo "$CMD" > $OUT/command.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o write -- $CMD > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_sqa -o sqa -- $CMD > $OUT/pmc_sqa.log 2>&1
cd $REPO
tail -1 $OUT/stats.log
