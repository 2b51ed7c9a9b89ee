#!/bin/bash
# rocprofv3 kernel stats + HBM / SQ counters for the lockstep kernel k_step (run on the GPU box through gpurun).
#   bash tools/profile_lockstep.sh [B A T N]  ->  gpurun_out/prof_lockstep_B<B>_<A>A<T>T/
set -u
REPO=$(pwd)
export TMPDIR=/tmp
B=${1:-65536}; A=${2:-20}; T=${3:-50}; N=${4:-40}
OUT=$REPO/gpurun_out/prof_lockstep_B${B}_${A}A${T}T
rm -rf $OUT; mkdir -p $OUT
CMD="python3 $REPO/tools/lockstep_probe.py $B $A $T $N"
echo "$CMD" > $OUT/command.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o write -- $CMD > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_sqa -o sqa -- $CMD > $OUT/pmc_sqa.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sqb -o sqb -- $CMD > $OUT/pmc_sqb.log 2>&1
cd $REPO
tail -1 $OUT/stats.log
