#!/bin/bash
# The round's contract lines (configs 2 / 4 / 5 + the two extra replay schedules), run AFTER tools/collect_all.sh has put the
# round's counters into profiles/counters.json, so that every line carries its roofline.   gpurun -- bash tools/bench_lines.sh r04
R=${1:-r05}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 > gpurun_out/${R}_bench.json 2> gpurun_out/${R}_bench.err
python bench.py --config 4 --steps 10 --warmup 2 > gpurun_out/${R}_bench_config4.json 2>> gpurun_out/${R}_bench.err
python bench.py --config 5 --steps 10 --warmup 2 > gpurun_out/${R}_bench_config5.json 2>> gpurun_out/${R}_bench.err
python bench.py --config 5 --steps 10 --warmup 2 --visibility 100,100,10,500 > gpurun_out/${R}_bench_config5_generalised.json 2>> gpurun_out/${R}_bench.err
python bench.py --config 5 --steps 10 --warmup 2 --visibility static > gpurun_out/${R}_bench_config5_static.json 2>> gpurun_out/${R}_bench.err
timeout 1500 python bench_configs.py > gpurun_out/${R}_bench_configs.jsonl 2>> gpurun_out/${R}_bench.err
tail -c 200 gpurun_out/${R}_bench.json
