#!/bin/bash
# GPU box: counter-unit calibration per instruction class -> gpurun_out/calib/{table.txt, summary.csv, ...}
# (copy table.txt + summary.csv into profiles/<round>_calib/).  Separate rocprofv3 runs, --pmc never combined with other domains.
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out/calib
rm -rf $OUT; mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/calib/pmc_calib.hip -o /tmp/pmc_calib || exit 1
cd /tmp
/tmp/pmc_calib > $OUT/table.txt 2>&1          # un-profiled: s_memtime clocks per instruction at 8 and 4 waves per SIMD
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU --kernel-trace --output-format csv -d $OUT/a -o a -- /tmp/pmc_calib > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d $OUT/b -o b -- /tmp/pmc_calib > $OUT/b.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 --kernel-trace --output-format csv -d $OUT/c -o c -- /tmp/pmc_calib > $OUT/c.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $OUT/d -o d -- /tmp/pmc_calib > $OUT/d.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s -o s -- /tmp/pmc_calib > $OUT/s.log 2>&1
rocprofv3 -L > $OUT/counters_available.txt 2>&1
cd $REPO
python3 - <<'PY'
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/calib/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
ops = ["v_fma_f64", "v_add_f64", "v_min_f64", "v_cvt_f32_f64", "v_fma_f32", "v_add_u32", "v_cndmask_b32", "v_mov_b32", "v_cmp_lt_f64",
       "v_mov_b32_dpp", "v_mbcnt_lo", "v_readlane_b32", "s_add_u32", "s_mul_i32", "ds_read_b64", "ds_read_b32", "v_cndmask_e64_sgpr",
       "v_cndmask+v_add_u32", "v_cmp+v_cndmask", "v_mov_b64", "v_min_u32_dpp", "v_cmp_lt_u32_e64", "v_cmp_lt_f64_e64", "v_bfe_u32", "v_lshlrev_b64"]
cols = sorted({c for d in agg.values() for c in d})
with open("gpurun_out/calib/summary.csv", "w") as f:
    f.write("kernel,op,lanes,grid,insts_per_wave," + ",".join(c + "_per_wave_inst" for c in cols) + "\n")
    for (k, grid), d in sorted(agg.items()):
        m = re.search(r"k_calib<(\d+), (\d+)>", k)
        if not m:
            continue
        waves = int(grid) // 64
        n = 65536.0 * waves            # instructions of the class in the whole launch
        f.write(f"\"{k}\",{ops[int(m.group(1))]},{m.group(2)},{grid},65536," +
                ",".join("%.4f" % (sum(d[c]) / len(d[c]) / n) if c in d else "" for c in cols) + "\n")
print(open("gpurun_out/calib/table.txt").read())
print(open("gpurun_out/calib/summary.csv").read())
PY
grep -i "SQ_INSTS_VALU\|SQ_ACTIVE_INST\|SQ_INST_CYCLES" $OUT/counters_available.txt | sort -u | head -60
