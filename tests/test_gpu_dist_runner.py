"""-m gpu: the runner level as one process per GPU (dcmrta_amd/dist_runner.py, examples/train_sharded.py) -- here
two ranks sharing the one GPU of the test box over gloo (DCM_FORCE_DEVICE=0; on a multi-GPU node the same command runs over
RCCL).  Checks: weights broadcast from rank 0, disjoint contiguous env shards whose union is the round's budget, one gathered
return vector identical on every rank, every recorded episode of every rank replayed bit-exactly through the oracle, the greedy
twins equal to the unsharded single-process job on the same instances, identical weights on all ranks after each
(all-reduced-gradient) optimizer step; and the reference's single-learner variant (experience gathered to rank 0)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(tmp, port, *extra):
    env = dict(os.environ, DCM_FORCE_DEVICE="0", DCM_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "examples", "train_sharded.py"), "--total-envs", "13", "--rounds", "2",
           "--embedding", "32", "--agents", "6", "9", "--tasks", "8", "12", "--dump", str(tmp), *extra]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    return [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]


def test_two_rank_sharded_training_rounds(gpu_device, oracle_lib, tmp_path):
    from dcmrta_amd.choice import env_seeds
    from dcmrta_amd.instances import generate_batch
    from dcmrta_amd.policy import AttentionNet
    from dcmrta_amd.runner import BatchedRunner
    from test_gpu_runner import _replay_recorded
    lines = _launch(tmp_path, 29581)
    assert len(lines) == 2 and all(l["decisions"] > 13 * 5 for l in lines)
    logs = [json.load(open(tmp_path / f"log_rank{r}.json")) for r in range(2)]
    for a, b in zip(*logs):
        assert a["weights_checksum"] == b["weights_checksum"]          # same Adam step on every rank (all-reduced gradients)
    assert logs[0][0]["weights_checksum"] != logs[0][1]["weights_checksum"]   # ... and the weights did move
    for rnd in range(2):
        d = [np.load(tmp_path / f"round{rnd}_rank{r}.npz") for r in range(2)]
        A, T = int(d[0]["A"]), int(d[0]["T"])
        assert (int(d[1]["A"]), int(d[1]["T"])) == (A, T)                               # the round's shape, drawn on rank 0
        assert (int(d[0]["lo"]), int(d[0]["hi"]), int(d[1]["lo"]), int(d[1]["hi"])) == (0, 7, 7, 13)
        full = np.concatenate([d[0]["summary"][:, 0], d[1]["summary"][:, 0]])
        for r in range(2):
            assert np.array_equal(d[r]["returns"], full)                               # every rank holds the whole return vector
        for r in range(2):
            lo, hi = int(d[r]["lo"]), int(d[r]["hi"])
            first = rnd * 13 + lo                                                       # round e plays instances [13 e, 13 e + 13)
            inst = generate_batch(hi - lo, A, T, base_seed=7, first=first)
            seeds = env_seeds(7, first, hi - lo)
            rec = {k: torch.from_numpy(d[r][k]) for k in ("agents", "tasks", "mask", "action", "leader", "active")}
            _replay_recorded(oracle_lib, rec, d[r]["summary"], inst, seeds, A, T)
        if rnd == 0:
            # the unsharded job with rank 0's initial weights: same instances, same greedy twins (deterministic argmax rollouts)
            torch.manual_seed(1234)
            single = BatchedRunner(n_envs=13, device=gpu_device, net_factory=lambda: AttentionNet(6, 5, 32), base_seed=7)
            w = single.get_weights()
            single.job(w, w, 0, A, T)
            ref = single.last["greedy_summary"].cpu().numpy()
            got = np.concatenate([d[0]["greedy_summary"], d[1]["greedy_summary"]])
            assert np.array_equal(got, ref)
            single.close()


def test_rank0_learner_variant(gpu_device, tmp_path):
    """--learner rank0: the experience of both ranks is gathered to rank 0 (driver.py's single learner), which steps and
    re-broadcasts; the ranks hold identical weights afterwards."""
    lines = _launch(tmp_path, 29583, "--learner", "rank0")
    assert len(lines) == 2 and lines[0]["decisions"] > 13 * 5
    logs = [json.load(open(tmp_path / f"log_rank{r}.json")) for r in range(2)]
    for a, b in zip(*logs):
        assert a["weights_checksum"] == b["weights_checksum"]
    assert logs[0][0]["weights_checksum"] != logs[0][1]["weights_checksum"]


def test_single_rank_round_at_a_compacting_batch_size(gpu_device):
    """One rank, a batch large enough for the runner's automatic policy compaction (>= 1024 envs: one graph per bucket size) and a
    learner that walks the round's decisions in minibatches with accumulated gradients: a plain single-GPU training round."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "train_sharded.py"), "--total-envs", "1100", "--rounds", "2",
                          "--embedding", "32", "--agents", "8", "10", "--tasks", "12", "16", "--minibatch", "4096"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 2 and all(l["decisions"] > 1100 * 8 and 0.0 < l["success_rate"] <= 1.0 for l in lines)
    assert lines[0]["weights_checksum"] != lines[1]["weights_checksum"]        # the optimizer step moved the weights
