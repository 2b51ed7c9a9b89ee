set -x
mkdir -p gpurun_out/r02a
python bench.py --steps 10 --warmup 3 > gpurun_out/r02a/bench_default.json 2> gpurun_out/r02a/bench_default.err
python bench.py --steps 10 --warmup 3 --tasks 49 --no-cpu-baseline --no-lockstep-probe > gpurun_out/r02a/bench_20_49.json 2>&1
python bench.py --steps 10 --warmup 3 --agents 15 --tasks 35 --no-cpu-baseline --no-lockstep-probe > gpurun_out/r02a/bench_15_35.json 2>&1
python bench.py --steps 10 --warmup 3 --envs 16384 --no-cpu-baseline --no-lockstep-probe > gpurun_out/r02a/bench_16k.json 2>&1
bash tools/pmc_probe.sh > gpurun_out/r02a/pmc_probe.log 2>&1
rocminfo | grep -i -E "compute unit|max clock|simd" | head > gpurun_out/r02a/rocminfo.txt
cat gpurun_out/r02a/*.json | cut -c1-400
tail -5 gpurun_out/r02a/pmc_probe.log
