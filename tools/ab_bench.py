#!/usr/bin/env python3
"""A/B already-built libraries against the PRODUCT line of bench.py on the GPU box (developer tool): interleaved rounds of
`bench.py --steps 20 --warmup 5` with its calibrated sub-batch streams (tools/ab.py times one launch on one stream instead).

    python tools/ab_bench.py name1=path1.so name2=path2.so ...   [env AB_ARGS="--config 4"]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = [a.split("=", 1) for a in sys.argv[1:]]
res = {n: [] for n, _ in libs}
for rnd in range(3):
    for name, so in libs:
        o = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline",
                            "--no-lockstep-probe", "--no-other-configs"] + os.environ.get("AB_ARGS", "").split(),
                           env=dict(os.environ, DCMRTA_HIP_LIB=os.path.abspath(so)), capture_output=True, text=True, timeout=900)
        line = [l for l in o.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(name, "FAILED", o.stderr[-300:])
            continue
        d = json.loads(line[-1])
        res[name].append((d["value"], d["config"].get("streams_per_gpu")))
for name, v in res.items():
    print(f"{name:12s} " + " ".join(f"{x[0] / 1e9:.4f}" for x in v) + "  e9 steps/s, streams " + str(sorted({x[1] for x in v})))
