set -x
mkdir -p gpurun_out/r02b
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02b/pytest.log 2>&1; echo "pytest rc=$?" 
tail -15 gpurun_out/r02b/pytest.log
for cfg in "" "--tasks 49" "--agents 15 --tasks 35" "--agents 10 --tasks 20"; do
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-lockstep-probe $cfg 2>/dev/null | cut -c1-120
done
