#!/bin/bash
# condense the gpurun_out/prof_* directories of the round's evidence run (tools/_sNN.sh) into profiles/ (run in the build container)
set -e
python tools/collect_profile.py r03_streams4 r03_streams4 > /dev/null
python tools/collect_profile.py r03_streams1 r03_streams1 > /dev/null      # last: the source of config 2's counters.json entry
python tools/collect_profile.py r03_c4 r03_config4 k_rollout_random > /dev/null
python tools/collect_profile.py r03_c5 r03_config5 k_replay > /dev/null
python tools/collect_profile.py r03_c5gen r03_config5_generalised k_replay > /dev/null
python tools/collect_profile.py r03_c5static r03_config5_static k_replay > /dev/null
python tools/collect_profile.py r03_15A35T r03_15A35T k_rollout_random > /dev/null
python tools/collect_lockstep.py r03_lockstep 4096 20 50 > /dev/null
python tools/collect_lockstep.py r03_lockstep 65536 20 50 > /dev/null
python tools/collect_lockstep.py r03_lockstep 16384 50 200 > /dev/null
python - <<'P'
import json, csv
c = json.load(open('profiles/counters.json'))
print({v.get('build_id') for v in c.values()})
for k, v in c.items():
    if 'SQ_INSTS_LDS_per_decision' in v:
        print(k, 'VALU %.1f SALU %.1f LDS %.1f HBM B/step %.1f' % (v['SQ_INSTS_VALU_per_decision'], v['SQ_INSTS_SALU_per_decision'], v['SQ_INSTS_LDS_per_decision'], v['hbm_bytes_per_decision']))
for f in ['r03_lockstep/k_step_B4096_20A50T.csv', 'r03_lockstep/k_step_B65536_20A50T.csv', 'r03_lockstep/k_step_B16384_50A200T.csv', 'r03_config4/kernel_stats.csv',
          'r03_config5/kernel_stats.csv', 'r03_config5_generalised/kernel_stats.csv', 'r03_config5_static/kernel_stats.csv', 'r03_streams1/kernel_stats.csv',
          'r03_streams4/kernel_stats.csv', 'r03_15A35T/kernel_stats.csv']:
    for l in open('profiles/' + f):
        if 'k_step' in l or 'k_rollout' in l or 'k_replay' in l:
            r = next(csv.reader([l])); print(f, 'calls', r[1], 'avg_us %.1f' % (float(r[3]) / 1e3))
P
