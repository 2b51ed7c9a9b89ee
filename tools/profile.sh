#!/bin/bash
# Profiling recipe behind profiles/ (run on the GPU box through gpurun; separate rocprofv3 runs, --pmc never combined with
# other trace domains).   bash tools/profile.sh <tag> [bench args...]  ->  gpurun_out/prof_<tag>/{stats,pmc_*}
# BENCH_PY=<script> selects another contract-line script (default bench.py).
set -u
TAG=${1:-r03}; shift || true
REPO=$(pwd)
export TMPDIR=/tmp
# the sub-batch streams of bench.py need 8 hardware queues; bench.py sets this itself, but under rocprofv3 the profiler's
# preloaded library initialises HIP before the script runs, so it has to be in the environment already
export GPU_MAX_HW_QUEUES=8
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
BENCH="python3 $REPO/${BENCH_PY:-bench.py} --steps 11 --warmup 1 --no-cpu-baseline --no-lockstep-probe $*"
echo "$BENCH" > $OUT/command.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $BENCH > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o write -- $BENCH > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_sqa -o sqa -- $BENCH > $OUT/pmc_sqa.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sqb -o sqb -- $BENCH > $OUT/pmc_sqb.log 2>&1
# instruction mix by class (priced per class with profiles/*_calib: fp64 ops take 4 clocks of a SIMD, 32-bit ops fewer)
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 --kernel-trace --output-format csv -d $OUT/pmc_cls -o cls -- $BENCH > $OUT/pmc_cls.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace --output-format csv -d $OUT/pmc_cls2 -o cls2 -- $BENCH > $OUT/pmc_cls2.log 2>&1
cd $REPO
grep -h '^{' $OUT/stats.log | tail -1 | cut -c1-300
