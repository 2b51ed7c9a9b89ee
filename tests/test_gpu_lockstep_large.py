"""-m gpu: the lockstep kernel at machine-filling batch sizes.  For grids of at least 8192 envs k_step builds the observation rows
in LDS (on top of the member-slot section it has already written back) and stores them as contiguous runs; smaller batches --
all the golden / oracle lockstep tests -- use the direct row-per-lane writer.  Here the staged writer is compared, after every
step, with (i) k_observe on the stepped state (direct writer) and (ii) the same instances stepped in a batch below the
threshold; a sample of envs is also replayed through the oracle.  Uniform, ragged and auto-reset batches, a one-chunk and a
multi-chunk shape (worker.py:57-76 is what one step + observation restates)."""
import numpy as np
import pytest
import torch

import oracle
from dcmrta_amd.batched_env import BatchedTaskEnv
from dcmrta_amd.choice import env_seeds
from dcmrta_amd.instances import generate_batch

pytestmark = pytest.mark.gpu
BIG = 8192          # k_step's staging threshold (dcmrta_env.hip: gridDim.x >= 8192)
SMALL = 1024


def _policy(obs, gen):
    """Valid random actions drawn on the device from the mask (same stream for every batch that uses the same generator state)."""
    w = (~obs.mask).float()
    w[~obs.active] = 0
    w[:, 0] += (w.sum(1) == 0).float()          # finished envs: any action, ignored
    return torch.multinomial(w, 1, generator=gen).squeeze(1).int()


def _snap(obs):
    return [x.clone() for x in (obs.agents, obs.tasks, obs.mask, obs.leader, obs.active)]


def _same(a, b, rows=None):
    for x, y in zip(a, b):
        if rows is not None:
            x, y = x[rows], y[rows]
        if x.dtype in (torch.float32, torch.float64):
            if not torch.equal(x.view(torch.int32 if x.dtype == torch.float32 else torch.int64),
                               y.view(torch.int32 if y.dtype == torch.float32 else torch.int64)):
                return False
        elif not torch.equal(x, y):
            return False
    return True


@pytest.mark.parametrize("A,T,steps,ragged,auto", [(5, 8, 60, False, False), (20, 50, 40, False, False), (20, 50, 40, True, False),
                                                    (5, 8, 120, False, True), (50, 200, 12, False, False)],
                         ids=["5A8T", "20A50T", "20A50T-ragged", "5A8T-autoreset", "50A200T"])
def test_staged_observation_writer_matches_direct_writer(gpu_device, A, T, steps, ragged, auto):
    inst = generate_batch(BIG, A, T, base_seed=77)
    if ragged:
        rng = np.random.default_rng(5)
        inst["n_agents"] = rng.integers(1, A + 1, BIG).astype(np.int32)
        inst["n_tasks"] = rng.integers(1, T + 1, BIG).astype(np.int32)
    seeds = env_seeds(9, 0, BIG)
    sub = {k: v[:SMALL] for k, v in inst.items()}
    big = BatchedTaskEnv(BIG, A, T, device=gpu_device, auto_reset=auto).load_instances(**inst)
    small = BatchedTaskEnv(SMALL, A, T, device=gpu_device, auto_reset=auto).load_instances(**sub)
    ob, os_ = big.reset(seeds), small.reset(seeds[:SMALL])
    gen = torch.Generator(device=gpu_device)
    for s in range(steps):
        gen.manual_seed(1000 + s)
        act = _policy(ob, gen)
        ob = big.step(act)
        staged = _snap(ob)
        direct = _snap(big.observe())                         # k_observe on the stepped state: direct row stores
        assert _same(staged, direct), f"step {s}: staged observation rows differ from k_observe's"
        os_ = small.step(act[:SMALL])                         # same instances / actions below the staging threshold
        assert _same(staged, _snap(os_), rows=slice(0, SMALL)), f"step {s}: batch of {BIG} differs from batch of {SMALL}"
    if auto:
        assert int(big.episodes().max()) >= 1                 # the auto-reset path ran under the staged writer
    big.close(); small.close()


def test_large_batch_lockstep_against_oracle(gpu_device, oracle_lib):
    """8192 x 20A/50T through dcm_step with injected oracle choices on a sample of envs: every per-decision output of the staged
    writer equals the oracle's record of that decision."""
    A, T, n_check = 20, 50, 24
    inst = generate_batch(BIG, A, T, base_seed=123)
    seeds = env_seeds(4, 0, BIG)
    idx = np.linspace(0, BIG - 1, n_check).astype(int)        # spread over the XCD-contiguous env blocks
    refs = []
    for b in idx:
        o = oracle.OracleEnv(A, T).load(inst["depot"][b], inst["task_xy"][b], inst["req"][b], inst["dur"][b])
        refs.append(o.rollout(int(seeds[b]), 0, oracle.POLICY_RANDOM, cap_steps=400, record=True))
    env = BatchedTaskEnv(BIG, A, T, device=gpu_device).load_instances(**inst)
    obs = env.reset(seeds)
    n_steps = min(len(r["action"]) for r in refs)
    gen = torch.Generator(device=gpu_device)
    for s in range(min(n_steps, 60)):
        gen.manual_seed(s)
        act = _policy(obs, gen)
        for j, b in enumerate(idx):                           # the sampled envs follow the oracle's (protocol) actions
            act[b] = int(refs[j]["action"][s])
        ag, tk, mk, ld = obs.agents.cpu().numpy(), obs.tasks.cpu().numpy(), obs.mask.cpu().numpy(), obs.leader.cpu().numpy()
        for j, b in enumerate(idx):
            r = refs[j]
            assert ld[b] == r["leader"][s]
            assert np.array_equal(ag[b].view(np.int32), np.ascontiguousarray(r["agents_obs"][s], np.float32).view(np.int32))
            assert np.array_equal(tk[b].view(np.int32), np.ascontiguousarray(r["tasks_obs"][s], np.float32).view(np.int32))
            assert np.array_equal(mk[b], r["mask"][s].astype(bool))
        obs = env.step(act)
    env.close()
