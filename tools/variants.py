#!/usr/bin/env python3
"""A/B different compile-time variants of the kernels on the GPU box (developer tool).

    python tools/variants.py "name1:-DFLAG1 -DFLAG2" "name2:" ...

Each variant is compiled to tools/_variants/lib_<name>.so and benchmarked with bench.py (interleaved, 2 rounds)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "dcmrta_amd", "csrc")
out = os.path.join(ROOT, "tools", "_variants")
os.makedirs(out, exist_ok=True)
variants = []
for spec in sys.argv[1:]:
    name, _, flags = spec.partition(":")
    so = os.path.join(out, f"lib_{name}.so")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-mllvm", "-phi-elim-split-all-critical-edges=1",
           "-shared"] + flags.split() + [os.path.join(src, "dcmrta_env.hip"), os.path.join(src, "dcmrta_replay.hip"), "-o", so]
    subprocess.check_call(cmd)
    variants.append((name, so))
res = {n: [] for n, _ in variants}
for rnd in range(2):
    for name, so in variants:
        env = dict(os.environ, DCMRTA_HIP_LIB=so)
        o = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "3", "--no-cpu-baseline",
                            "--no-lockstep-probe"] + os.environ.get("VARIANTS_ARGS", "").split(),
                           env=env, capture_output=True, text=True)
        line = [l for l in o.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(name, "FAILED", o.stderr[-400:])
            continue
        j = json.loads(line[-1])
        res[name].append(j["roofline"]["avg_launch_ms"])
for name, v in res.items():
    print(f"{name:24s} launch ms: " + " ".join(f"{x:.4f}" for x in v))
if os.environ.get("VARIANTS_PMC"):   # per-decision instruction counts of k_rollout_random for every variant
    import collections
    import csv
    import glob
    import shutil
    for name, so in variants:
        d = os.path.join(out, f"pmc_{name}")
        shutil.rmtree(d, ignore_errors=True)
        env = dict(os.environ, DCMRTA_HIP_LIB=so, TMPDIR="/tmp")
        o = subprocess.run(["rocprofv3", "--pmc", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY",
                            "SQ_WAIT_INST_ANY", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_WR", "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "x", "--",
                            "python3", os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "1", "--no-cpu-baseline",
                            "--no-lockstep-probe", "--streams", "1"],
                           env=env, capture_output=True, text=True, cwd="/tmp")
        line = [l for l in o.stdout.splitlines() if l.startswith("{")]
        cfg = json.loads(line[-1])["config"] if line else None
        decisions = (cfg["decisions_per_step_per_gpu"] * 5 + cfg["decisions_in_warmup_per_gpu"]) if cfg else float("nan")
        agg = collections.defaultdict(list)
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_rollout" in r["Kernel_Name"]:
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(name, {k: round(sum(v) / decisions, 1) for k, v in sorted(agg.items())})
