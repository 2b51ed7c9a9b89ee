#!/usr/bin/env python3
"""Per-decision instruction counters of the persistent rollout kernel for already-built libraries (developer tool, GPU box).

    python tools/pmc_lib.py name=lib.so ...      [env PMC_ARGS="--agents 15 --tasks 35"]

Two rocprofv3 --pmc passes per library over `bench.py --streams 1` (counters only, no other trace domain)."""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "pmc_lib")
GROUPS = [["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_WR", "SQ_INSTS_VMEM_RD", "SQ_WAVES"],
          ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"]]
for spec in sys.argv[1:]:
    name, so = spec.split("=", 1)
    tot = {}
    for gi, grp in enumerate(GROUPS):
        d = os.path.join(OUT, f"{name}_{gi}")
        shutil.rmtree(d, ignore_errors=True)
        env = dict(os.environ, DCMRTA_HIP_LIB=os.path.abspath(so), TMPDIR="/tmp", GPU_MAX_HW_QUEUES="8")
        o = subprocess.run(["rocprofv3", "--pmc"] + grp + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", "x", "--",
                            "python3", os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "1", "--no-cpu-baseline",
                            "--no-lockstep-probe", "--streams", "1"] + os.environ.get("PMC_ARGS", "").split(),
                           env=env, capture_output=True, text=True, cwd="/tmp")
        line = [l for l in o.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(name, "FAILED", o.stderr[-600:])
            continue
        cfg = json.loads(line[-1])["config"]
        decisions = cfg["decisions_per_step_per_gpu"] * 5 + cfg["decisions_in_warmup_per_gpu"]
        agg = collections.defaultdict(float)
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_rollout" in r["Kernel_Name"]:
                    agg[r["Counter_Name"]] += float(r["Counter_Value"])
        tot.update({k: v / decisions for k, v in agg.items()})
    print(name, json.dumps({k: round(v, 1) for k, v in sorted(tot.items())}))
