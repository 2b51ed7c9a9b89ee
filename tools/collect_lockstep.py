#!/usr/bin/env python3
"""Condense gpurun_out/prof_lockstep (tools/profile_lockstep.sh) into profiles/<name>/k_step_*.csv and profiles/counters.json.

    python tools/collect_lockstep.py <name> <B> <A> <T>
"""
import collections
import csv
import glob
import json
import os
import sys

name, B, A, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
base = f"gpurun_out/prof_lockstep_B{B}_{A}A{T}T/"
dst = f"profiles/{name}"
os.makedirs(dst, exist_ok=True)
cmd = open(base + "command.txt").read().strip()
stats = glob.glob(base + "stats/**/*kernel_stats.csv", recursive=True)[0]
out = f"{dst}/k_step_B{B}_{A}A{T}T.csv"
avg_ns = None
with open(out, "w") as f:
    f.write(f"# {cmd}\n")
    for l in open(stats):
        if l.startswith('"Name"') or "k_step" in l:
            f.write(l)
            if "k_step" in l:
                avg_ns = float(next(csv.reader([l]))[3])
    tot = {}
    f.write("counter,dispatches,avg_per_dispatch\n")
    for d in ("pmc_fetch", "pmc_write", "pmc_sqa", "pmc_sqb"):
        agg = collections.defaultdict(list)
        hits = glob.glob(base + f"{d}/**/*counter_collection.csv", recursive=True)
        if not hits:
            continue
        for r in csv.DictReader(open(hits[0])):
            if "k_step" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            f.write("%s,%d,%.3f\n" % (k, len(v), sum(v) / len(v)))
            tot[k] = sum(v) / len(v)
W = 2 * (64 + 48 * A + 96 * T) + 24 * A + 21 * (T + 1) + 4
traffic = (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024
path = "profiles/counters.json"
allc = json.load(open(path)) if os.path.exists(path) else {}
build_id = None
for l in open(base + "stats.log"):
    if l.startswith("build_id "):
        build_id = l.split()[1]
allc[f"k_step:{B}x{A}A{T}T"] = dict(source=out, avg_launch_us=avg_ns / 1e3, hbm_bytes_per_launch=traffic, build_id=build_id,
                                    algorithmic_bytes_per_launch=B * W, traffic_over_algorithmic=traffic / (B * W),
                                    hbm_frac_rocprof=B * W / (avg_ns * 1e-9) / 8e12,
                                    **{k + "_per_launch": v for k, v in tot.items()})
json.dump(allc, open(path, "w"), indent=1, sort_keys=True)
print(json.dumps(allc[f"k_step:{B}x{A}A{T}T"], indent=1))
