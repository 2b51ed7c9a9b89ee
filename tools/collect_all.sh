#!/bin/bash
# condense the gpurun_out/prof_* directories of the round's evidence run (tools/evidence_run.sh) into profiles/ (run in the build container)
set -e
R=${1:-r05}
python tools/collect_profile.py ${R}_s1 ${R}_streams1 k_rollout_fast > /dev/null        # one 4096-env launch per pass: the PRIO instantiation (wave priorities)
python tools/collect_profile.py ${R}_s4 ${R}_streams4 k_rollout_fast > /dev/null        # last = the source of config 2's counters.json entry: the product line's four 1024-env launches
python tools/collect_profile.py ${R}_c4 ${R}_config4 k_rollout_fast_mc > /dev/null
python tools/collect_profile.py ${R}_c5 ${R}_config5 k_replay_fast > /dev/null
python tools/collect_profile.py ${R}_c5gen ${R}_config5_generalised k_replay > /dev/null
python tools/collect_profile.py ${R}_c5static ${R}_config5_static k_replay > /dev/null
python tools/collect_profile.py ${R}_15A35T ${R}_15A35T k_rollout_fast > /dev/null
python tools/collect_profile.py ${R}_70A130T ${R}_70A130T k_rollout_fast_g > /dev/null
python tools/collect_lockstep.py ${R}_lockstep 4096 20 50 > /dev/null
python tools/collect_lockstep.py ${R}_lockstep 65536 20 50 > /dev/null
python tools/collect_lockstep.py ${R}_lockstep 16384 50 200 > /dev/null
# the steady-state collection loop: the probe's log + the kernel stats of the same loop (k_step_fast, k_terminal_flush)
[ -s gpurun_out/${R}_lockstep_steady.log ] && grep -v "amdgpu.ids" gpurun_out/${R}_lockstep_steady.log > profiles/${R}_lockstep_steady.log
if [ -d gpurun_out/prof_${R}_steady ]; then
  mkdir -p profiles/${R}_lockstep_steady
  cp gpurun_out/prof_${R}_steady/command.txt profiles/${R}_lockstep_steady/
  f=$(find gpurun_out/prof_${R}_steady -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && (head -1 $f; grep -E "k_step|k_terminal" $f) > profiles/${R}_lockstep_steady/kernel_stats.csv
  # ... and into counters.json (bench.py: lockstep_kernel.steady_state_rocprof_avg_us)
  python - ${R} <<'P'
import csv, json, sys
R = sys.argv[1]
rows = {("flush" if "k_terminal" in r["Name"] else "step"): r for r in csv.DictReader(open(f"profiles/{R}_lockstep_steady/kernel_stats.csv"))}
bid = [l.split()[1] for l in open(f"gpurun_out/prof_{R}_steady/stats.log") if l.startswith("build_id ")]
c = json.load(open("profiles/counters.json"))
c["k_step_steady:4096x20A50T"] = dict(source=f"profiles/{R}_lockstep_steady/kernel_stats.csv", build_id=bid[-1] if bid else None,
                                      avg_launch_us=float(rows["step"]["AverageNs"]) / 1e3, launches=int(rows["step"]["Calls"]),
                                      flush_avg_us=float(rows["flush"]["AverageNs"]) / 1e3 if "flush" in rows else None,
                                      flush_launches=int(rows["flush"]["Calls"]) if "flush" in rows else 0)
json.dump(c, open("profiles/counters.json", "w"), indent=1, sort_keys=True)
P
fi
for f in bench bench_config4 bench_config5 bench_config5_generalised bench_config5_static; do
  [ -s gpurun_out/${R}_$f.json ] && grep '^{' gpurun_out/${R}_$f.json | tail -1 > profiles/${R}_$f.json
done
python - <<'P'
import json
c = json.load(open('profiles/counters.json'))
print({v.get('build_id') for v in c.values()})
for k, v in sorted(c.items()):
    if 'SQ_INSTS_LDS_per_decision' in v:
        print(k, 'VALU %.1f SALU %.1f LDS %.1f BRANCH %.1f HBM B/step %.1f' % (v['SQ_INSTS_VALU_per_decision'], v['SQ_INSTS_SALU_per_decision'], v['SQ_INSTS_LDS_per_decision'], v.get('SQ_INSTS_BRANCH_per_decision', float('nan')), v['hbm_bytes_per_decision']))
    elif 'avg_launch_us' in v:
        print(k, 'avg_launch_us %.1f traffic/algorithmic %.3f' % (v['avg_launch_us'], v.get('traffic_over_algorithmic', float('nan'))))
P
