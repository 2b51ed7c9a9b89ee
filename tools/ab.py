#!/usr/bin/env python3
"""A/B already-built libraries against bench.py on the GPU box (developer tool): interleaved rounds, one 4096-env launch per pass.

    python tools/ab.py name1=path1.so name2=path2.so ...   [env AB_ARGS="--agents 15 --tasks 35"]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = [a.split("=", 1) for a in sys.argv[1:]]
res = {n: [] for n, _ in libs}
for rnd in range(3):
    for name, so in libs:
        o = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "3", "--no-cpu-baseline",
                            "--no-lockstep-probe", "--streams", "1"] + os.environ.get("AB_ARGS", "").split(),
                           env=dict(os.environ, DCMRTA_HIP_LIB=os.path.abspath(so)), capture_output=True, text=True, timeout=600)
        line = [l for l in o.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(name, "FAILED", o.stderr[-400:])
            continue
        res[name].append(json.loads(line[-1])["roofline"]["avg_launch_ms"])
for name, v in res.items():
    print(f"{name:16s} launch ms: " + " ".join(f"{x:.4f}" for x in v) + (f"   mean {sum(v) / len(v):.4f}" if v else ""))
