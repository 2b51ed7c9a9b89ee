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
