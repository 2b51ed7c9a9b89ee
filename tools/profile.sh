#!/bin/bash
# Profiling recipe behind profiles/ (run on the GPU box through gpurun; separate rocprofv3 runs, --pmc never combined with
# other trace domains).   bash tools/profile.sh <tag> [bench args...]  ->  gpurun_out/prof_<tag>/{stats,pmc_*}
set -u
TAG=${1:-r02}; shift || true
REPO=$(pwd)
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
BENCH="python3 $REPO/bench.py --steps 11 --warmup 1 --no-cpu-baseline --no-lockstep-probe $*"
echo "$BENCH" > $OUT/command.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $BENCH > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o write -- $BENCH > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_sqa -o sqa -- $BENCH > $OUT/pmc_sqa.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sqb -o sqb -- $BENCH > $OUT/pmc_sqb.log 2>&1
cd $REPO
grep -h '^{' $OUT/stats.log | tail -1 | cut -c1-300
