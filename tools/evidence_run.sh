#!/bin/bash
# The round's evidence run (on the GPU box through gpurun): every rocprofv3 profile behind profiles/ + the contract lines.
#   gpurun --timeout 3000 -- bash tools/evidence_run.sh r04      then, in the build container:  bash tools/collect_all.sh r04
R=${1:-r05}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/profile.sh ${R}_s4 --streams 4
bash tools/profile.sh ${R}_s1 --streams 1
bash tools/profile.sh ${R}_15A35T --agents 15 --tasks 35 --streams 1
bash tools/profile.sh ${R}_70A130T --agents 70 --tasks 130 --streams 1
bash tools/profile.sh ${R}_c4 --config 4 --envs 8192 --streams 1
bash tools/profile.sh ${R}_c5 --config 5
bash tools/profile.sh ${R}_c5gen --config 5 --visibility 100,100,10,500
bash tools/profile.sh ${R}_c5static --config 5 --visibility static
bash tools/profile_lockstep.sh 4096 20 50 100
bash tools/profile_lockstep.sh 65536 20 50 100
bash tools/profile_lockstep.sh 16384 50 200 100
# the lockstep API in the steady state of a collection loop (auto-reset: deferred terminal metrics, restart image) + the kernel stats
# of the same loop at the BASELINE batch (k_step_fast and k_terminal_flush, one launch in 32)
python tools/lockstep_probe.py 4096 20 50 300 > /dev/null 2>&1          # (the box's first GPU work: clocks still ramping)
for B in 1024 4096 65536; do python tools/lockstep_probe.py $B 20 50 350 steady; python tools/lockstep_probe.py $B 20 50 100; done > gpurun_out/${R}_lockstep_steady.log 2>&1
( export TMPDIR=/tmp; O=$PWD/gpurun_out/prof_${R}_steady; rm -rf $O; mkdir -p $O; P=$PWD/tools/lockstep_probe.py; echo "python3 $P 4096 20 50 350 steady" > $O/command.txt; cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o stats -- python3 $P 4096 20 50 350 steady > $O/stats.log 2>&1 )
python bench.py --steps 20 --warmup 5 > gpurun_out/${R}_bench.json 2> gpurun_out/${R}_bench.err
python bench.py --config 4 --steps 10 --warmup 2 > gpurun_out/${R}_bench_config4.json 2>> gpurun_out/${R}_bench.err
python bench.py --config 5 --steps 10 --warmup 2 > gpurun_out/${R}_bench_config5.json 2>> gpurun_out/${R}_bench.err
python bench.py --config 5 --steps 10 --warmup 2 --visibility 100,100,10,500 > gpurun_out/${R}_bench_config5_generalised.json 2>> gpurun_out/${R}_bench.err
python bench.py --config 5 --steps 10 --warmup 2 --visibility static > gpurun_out/${R}_bench_config5_static.json 2>> gpurun_out/${R}_bench.err
# keep only what tools/collect_*.py read (gpurun copies back at most 64 MiB): the stats / counter CSVs, the logs, the commands
echo "gpurun_out before pruning: $(du -sh gpurun_out | cut -f1)"
find gpurun_out/prof_* -type f ! \( -name '*kernel_stats.csv' -o -name '*counter_collection.csv' -o -name '*.log' -o -name 'command.txt' \) -delete
find gpurun_out/prof_* -type f -name '*.log' -size +256k -exec sh -c 'tail -c 65536 "$1" > "$1.t" && mv "$1.t" "$1"' _ {} \;
# the counter CSVs hold one row per dispatch and counter of EVERY kernel of the process (torch's included): keep the env kernels' rows
for f in $(find gpurun_out/prof_* -type f -name '*counter_collection.csv'); do
  awk 'NR == 1 || /k_step|k_rollout|k_replay|k_observe|k_reset/' "$f" > "$f.t" && mv "$f.t" "$f"
done
du -a gpurun_out | sort -n | tail -5
echo "gpurun_out after pruning: $(du -sh gpurun_out | cut -f1)"
tail -c 300 gpurun_out/${R}_bench.json
