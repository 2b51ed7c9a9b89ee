#!/bin/bash
# Profiling recipe used for profiles/ (run on the GPU box through gpurun).
#   bash tools/profile.sh <tag>   ->  gpurun_out/prof_<tag>/{stats,pmc_fetch,pmc_write}
set -u
TAG=${1:-r01}
REPO=$(pwd)
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
BENCH="python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-lockstep-probe"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $BENCH > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o write -- $BENCH > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -o sq -- $BENCH > $OUT/pmc_sq.log 2>&1
cd $REPO
find $OUT -name "*.csv" | head -30
tail -3 $OUT/stats.log
