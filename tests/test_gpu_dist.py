"""-m gpu: the N > 1 launch path of bench.py end to end -- `python -m torch.distributed.run --nproc-per-node 2 bench.py
--gpus 2`, here as two ranks sharing the one GPU of the test box (DCM_FORCE_DEVICE=0) with the gloo backend (RCCL needs
one device per rank; on a multi-GPU node the same command runs over RCCL/xGMI).  Checks the contract line: aggregate
value over both ranks, weak scaling, the sharded env blocks and the return all-gather."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_bench_line(gpu_device):
    env = dict(os.environ, DCM_FORCE_DEVICE="0", DCM_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29571", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--envs", "512"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:] + out.stderr[-2000:]          # rank 0 prints exactly one JSON line
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["steps"] == 3 and j["warmup"] == 1
    assert j["metric"] == "env_steps_per_sec" and j["value"] > 0 and j["vs_baseline"] is None
    assert j["config"]["envs_per_gpu"] == 512 and "x2" in j["config"]["sharding"]
    # both ranks' decisions are counted: about 2 x 512 envs x 3 episodes x ~120 decisions per pass
    per_pass_all_ranks = j["value"] * j["ms_per_step"] / 1e3
    assert 2 * 512 * 3 * 80 < per_pass_all_ranks < 2 * 512 * 3 * 160
    assert "cpu_baseline" not in j                                           # rank 0 at N = 1 only


def test_single_gpu_bench_contract_line(gpu_device):
    """python bench.py (N = 1): one JSON line with the contract keys, the roofline object and the bounded CPU baseline."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-lockstep-probe"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:] + out.stderr[-2000:]
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["dtype"] == "f64" and j["data"] == "synthetic" and j["higher_is_better"] is True
    assert "workload" in j["config"] and "model" not in j["config"] and j["config"]["envs_per_gpu"] == 4096
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["traffic"] is None or r["traffic"] > 0
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "steps/s" and "sample" in c
    assert j["value"] > 1e6          # the north-star floor (1M env-steps/s on one MI355X)
