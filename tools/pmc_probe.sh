#!/bin/bash
# one-off counter probe of the rollout kernel (GPU box)
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_probe
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/a -o a -- python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-lockstep-probe > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_VSKIPPED GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/b -o b -- python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-lockstep-probe > $OUT/b.log 2>&1
cd $REPO
python3 - <<'PY'
import csv, glob, collections
for d in ("a","b"):
    agg=collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pmc_probe/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_rollout" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print({k: round(sum(v)/len(v)/1472143.6,1) for k,v in sorted(agg.items())})
PY
tail -2 $OUT/b.log | cut -c1-200
