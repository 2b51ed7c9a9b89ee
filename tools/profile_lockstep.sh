#!/bin/bash
# rocprofv3 kernel stats + HBM counters for the lockstep kernel k_step (run on the GPU box through gpurun).
set -u
REPO=$(pwd)
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/prof_lockstep
mkdir -p $OUT
CMD="python3 $REPO/tools/lockstep_probe.py 65536 20 50 40"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o write -- $CMD > $OUT/pmc_write.log 2>&1
cd $REPO
tail -1 $OUT/stats.log
