mkdir -p gpurun_out/r02d
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02d/pytest.log 2>&1; echo "pytest rc=$?" 
tail -15 gpurun_out/r02d/pytest.log
for cfg in "" "--agents 15 --tasks 35" "--envs 8192 --agents 50 --tasks 200 --episodes 1"; do
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-lockstep-probe $cfg 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['config']['workload'][:40], j['value'], j['roofline']['avg_launch_ms'])"
done
