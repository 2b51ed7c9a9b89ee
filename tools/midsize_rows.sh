#!/bin/bash
# Steps/s of the persistent rollout kernel over mid-size shapes (the class of rollout_fast_g.hpp) -> gpurun_out/<round>_midsize.jsonl
#   gpurun -- bash tools/midsize_rows.sh r04        (bench.py per shape: 4096 envs, 3 episodes per env per pass, auto streams; + 70A/130T at 8192)
R=${1:-r04}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/${R}_midsize.jsonl
: > $out
run() {
  python bench.py --agents $1 --tasks $2 --envs $3 --steps 8 --warmup 2 --no-cpu-baseline --no-lockstep-probe --no-other-configs $4 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.readline())
print(json.dumps({'shape': '$1A/$2T', 'envs': $3, 'streams': d['config']['streams_per_gpu'], 'kernel': d['roofline']['kernel'], 'steps_per_s': d['value'], 'ms_per_pass': d['ms_per_step'], 'decisions_per_pass': d['config']['decisions_per_step_per_gpu']}))" >> $out
}
for sh in "70 130" "65 65" "30 100" "100 64" "50 150" "64 192" "128 256"; do set -- $sh; run $1 $2 4096 ""; done
run 70 130 4096 "--streams 1"
run 70 130 8192 ""
cat $out
