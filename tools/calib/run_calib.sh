#!/bin/bash
# GPU box: counter-unit calibration -> gpurun_out/calib/{a,b}_counter_collection.csv (copy the summary into profiles/)
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out/calib
rm -rf $OUT; mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/calib/pmc_calib.hip -o /tmp/pmc_calib || exit 1
cd /tmp
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU --kernel-trace --output-format csv -d $OUT/a -o a -- /tmp/pmc_calib > $OUT/a.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/b -o b -- /tmp/pmc_calib > $OUT/b.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s -o s -- /tmp/pmc_calib > $OUT/s.log 2>&1
cd $REPO
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/calib/*/**/*counter_collection.csv", recursive=True) + glob.glob("gpurun_out/calib/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    print(k, {c: sum(v) / len(v) for c, v in sorted(d.items())})
for f in glob.glob("gpurun_out/calib/s/**/*kernel_stats.csv", recursive=True) + glob.glob("gpurun_out/calib/s/*kernel_stats.csv"):
    print(open(f).read())
PY
