#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag> (written by tools/profile.sh) into profiles/<name>/ and profiles/counters.json.

    python tools/collect_profile.py <tag> <name> [kernel-substring]

profiles/<name>/kernel_stats.csv            rocprofv3 --kernel-trace --stats summary of the bench command
profiles/<name>/pmc_<kernel>.csv            per-dispatch averages of every counter, number of dispatches
profiles/<name>/bench_line.json             the bench JSON line of the stats run (decision counts of the profiled passes)
profiles/counters.json                      "<kernel>:<A>A<T>T" -> per-decision averages that bench.py turns into roofline.frac
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag, name = sys.argv[1], sys.argv[2]
kern = sys.argv[3] if len(sys.argv) > 3 else "k_rollout_random"
base = f"gpurun_out/prof_{tag}/"
dst = f"profiles/{name}"
os.makedirs(dst, exist_ok=True)


def find(pattern):
    hits = glob.glob(base + pattern, recursive=True)
    if not hits:
        raise SystemExit(f"missing {base}{pattern}")
    return hits[0]


shutil.copy(find("stats/**/*kernel_stats.csv"), f"{dst}/kernel_stats.csv")
shutil.copy(base + "command.txt", f"{dst}/command.txt")
line = [l for l in open(base + "stats.log") if l.startswith("{")][-1]
bench = json.loads(line)
json.dump(bench, open(f"{dst}/bench_line.json", "w"), indent=1)
cfg = bench["config"]
passes = bench["steps"] + bench["warmup"]
decisions = cfg["decisions_per_step_per_gpu"] * bench["steps"] + cfg["decisions_in_warmup_per_gpu"]
rows, meta, tot = [], None, {}
for d in ("pmc_fetch", "pmc_write", "pmc_sqa", "pmc_sqb", "pmc_cls", "pmc_cls2"):
    agg = collections.defaultdict(list)
    hits = glob.glob(base + f"{d}/**/*counter_collection.csv", recursive=True)
    if not hits:
        if d.startswith("pmc_cls"):
            continue                      # instruction-class passes are optional
        raise SystemExit(f"missing {base}{d}")
    for r in csv.DictReader(open(hits[0])):
        if kern in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = (r["VGPR_Count"], r["SGPR_Count"], r["Scratch_Size"], r["Grid_Size"], r["Workgroup_Size"])
    for k, v in sorted(agg.items()):
        rows.append((k, len(v), sum(v) / len(v)))
        tot[k] = sum(v)
with open(f"{dst}/pmc_{kern}.csv", "w") as f:
    f.write("counter,dispatches,avg_per_dispatch,total\n")
    for k, n, avg in rows:
        f.write("%s,%d,%.3f,%.1f\n" % (k, n, avg, tot[k]))
    f.write("# VGPR,SGPR,Scratch,Grid,WG = %s\n" % (meta,))
    f.write("# command: %s\n" % open(base + "command.txt").read().strip())
    f.write("# passes %d (incl. warm-up), decisions in all passes %d\n" % (passes, decisions))
per = {f"{k}_per_decision": v / decisions for k, v in tot.items()}
vis = cfg.get("visibility")
vis_tag = ":static" if vis == "static" else ("" if not vis or list(vis) == [20, 20, 10, 100] else ":vis" + "-".join(str(v) for v in vis))
key = f"{kern}:{cfg['agents']}A{cfg['tasks']}T{vis_tag}"
entry = dict(per, source=f"profiles/{name}/pmc_{kern}.csv", envs_per_gpu=cfg["envs_per_gpu"], streams=cfg.get("streams_per_gpu", 1),
             decisions_profiled=decisions,
             # dcm_build_id of the library that was profiled: bench.py reports "stale" when it runs another build
             build_id=bench["roofline"].get("build_id"),
             hbm_bytes_per_decision=(2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 / decisions,
             note="FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM (16 B/lane coalesced record loads); SQ_WAVE_CYCLES / SQ_WAIT_* "
                  "in quad-cycles, SQ_ACTIVE_INST_VALU = instruction count (profiles/r03_calib); a 'decision' of k_replay is one "
                  "agent_step call")
path = "profiles/counters.json"
allc = json.load(open(path)) if os.path.exists(path) else {}
allc[key] = entry
json.dump(allc, open(path, "w"), indent=1, sort_keys=True)
for l in open(f"{dst}/kernel_stats.csv"):
    if kern in l:
        print(l.strip()[:220])
print(key, {k: round(v, 2) for k, v in entry.items() if isinstance(v, float)})
