#!/usr/bin/env python3
"""A/B already-built libraries on the lockstep kernel k_step (developer tool, GPU box): interleaved rounds of tools/lockstep_probe.py.

    python tools/ab_lockstep.py name1=path1.so name2=path2.so ...   [env AB_SHAPES="4096,20,50,80 65536,20,50,40"]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = [a.split("=", 1) for a in sys.argv[1:]]
shapes = [tuple(x.split(",")) for x in os.environ.get("AB_SHAPES", "4096,20,50,80 65536,20,50,40").split()]
res = {(n, s): [] for n, _ in libs for s in shapes}
for rnd in range(3):
    for name, so in libs:
        for shp in shapes:
            o = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "lockstep_probe.py"), *shp],
                               env=dict(os.environ, DCMRTA_HIP_LIB=os.path.abspath(so)), capture_output=True, text=True, timeout=600)
            m = re.search(r"median ([0-9.]+) us", o.stdout)
            if not m:
                print(name, shp, "FAILED", o.stderr[-300:])
                continue
            res[(name, shp)].append(float(m.group(1)))
for (name, shp), v in res.items():
    print(f"{name:12s} B={shp[0]:>6s} {shp[1]}A/{shp[2]}T  k_step median us: " + " ".join(f"{x:.1f}" for x in v))
