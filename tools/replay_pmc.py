#!/usr/bin/env python3
"""Per-agent-step instruction counts of k_replay on the GPU box (developer tool): two rocprofv3 --pmc passes over
bench.py --config 5 [extra args], printed per step.   python tools/replay_pmc.py [bench args...]"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GROUPS = [["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"],
          ["SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_INSTS_BRANCH", "SQ_WAVES"]]
if os.environ.get("REPLAY_PMC_CLASSES"):      # the VALU instruction classes as well (two more passes)
    GROUPS += [["SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU_CVT",
                "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64"],
               ["SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32"]]
out = {}
for gi, grp in enumerate(GROUPS):
    d = f"/tmp/replay_pmc_{gi}"
    shutil.rmtree(d, ignore_errors=True)
    o = subprocess.run(["rocprofv3", "--pmc", *grp, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "x", "--", "python3",
                        os.path.join(ROOT, "bench.py"), "--config", "5", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                        "--no-lockstep-probe"] + sys.argv[1:], env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, cwd="/tmp")
    line = [l for l in o.stdout.splitlines() if l.startswith("{")]
    if not line:
        print("FAILED", o.stderr[-600:]); sys.exit(1)
    j = json.loads(line[-1])
    cfg = j["config"]
    steps = cfg["decisions_per_step_per_gpu"] * 3 + cfg["decisions_in_warmup_per_gpu"]
    agg = collections.defaultdict(float)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_replay" in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in agg.items():
        out[k] = v / steps
    out["launch_ms"] = j["roofline"]["avg_launch_ms"]
print(json.dumps({k: round(v, 2) for k, v in sorted(out.items())}))
