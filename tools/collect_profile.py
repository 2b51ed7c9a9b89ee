#!/usr/bin/env python3
"""Post-process gpurun_out/prof_<tag> (written by tools/profile.sh) into profiles/<name>/ and profiles/pmc_traffic.json."""
import collections
import csv
import json
import os
import shutil
import sys

tag, name = sys.argv[1], sys.argv[2]
base = f"gpurun_out/prof_{tag}/"
dst = f"profiles/{name}"
os.makedirs(dst, exist_ok=True)
shutil.copy(base + "stats/stats_kernel_stats.csv", f"{dst}/kernel_stats.csv")
rows, meta = [], None
for f in ("pmc_fetch/fetch_counter_collection.csv", "pmc_write/write_counter_collection.csv", "pmc_sq/sq_counter_collection.csv"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(base + f)):
        if "k_rollout" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = (r["VGPR_Count"], r["SGPR_Count"], r["Scratch_Size"], r["Grid_Size"], r["Workgroup_Size"])
    for k, v in sorted(agg.items()):
        rows.append((k, len(v), sum(v) / len(v)))
with open(f"{dst}/pmc_k_rollout_random.csv", "w") as f:
    f.write("counter,dispatches,avg_per_dispatch\n")
    for r in rows:
        f.write("%s,%d,%.3f\n" % r)
    f.write("# VGPR,SGPR,Scratch,Grid,WG = %s\n" % (meta,))
    f.write("# command: python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline (3 episodes per env per launch)\n")
d = dict((r[0], r[2]) for r in rows)
traffic = (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024
json.dump({"workload": "4096x20A50T", "kernel": "k_rollout_random", "episodes_per_launch": 3, "hbm_bytes_per_launch": traffic,
           "fetch_size_kib": d["FETCH_SIZE"], "write_size_kib": d["WRITE_SIZE"],
           "note": f"separate --pmc passes (profiles/{name}); FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM (16 B/lane coalesced record "
                   "loads). WRITE_SIZE counts L2 write-backs incl. those absorbed by the Infinity Cache: with 4096 resident waves the "
                   "per-XCD footprint (512 records + their observation rows ~ 3.7 MB) sits at the 4 MiB L2 capacity, so repeatedly "
                   "rewritten observation lines get evicted."},
          open("profiles/pmc_traffic.json", "w"), indent=1)
for l in open(f"{dst}/kernel_stats.csv"):
    if "k_rollout" in l:
        print(l.strip()[:200])
print({k: round(v, 1) for k, v in d.items()}, "traffic/launch", traffic)
