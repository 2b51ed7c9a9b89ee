"""-m gpu: BatchedRunner.job keeps the reference runner's contract (runner.py:58-71, driver.py:135-176) and its
experience replays bit-exactly through the oracle (same choices -> same observations)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_job_contract_and_replay(gpu_device, oracle_lib):
    from dcmrta_amd.choice import env_seeds
    from dcmrta_amd.instances import generate_batch
    from dcmrta_amd.policy import AttentionNet
    from dcmrta_amd.runner import METRIC_KEYS, BatchedRunner
    torch.manual_seed(0)
    B, A, T = 6, 10, 20
    small = lambda: AttentionNet(6, 5, 32)
    r = BatchedRunner(metaAgentID=3, n_envs=B, device=gpu_device, net_factory=small, base_seed=11)
    w = {k: v.clone() for k, v in r.get_weights().items()}
    jobResults, metrics, info = r.job(w, w, episodeNumber=2, agents_num=A, tasks_num=T, as_lists=True)
    assert info == {"id": 3, "episode_number": 2} and set(metrics) == set(METRIC_KEYS)
    assert len(jobResults) == 9 and jobResults[7] == [] and jobResults[8] == []
    # driver.py:158-164 stacks the per-decision tensors
    agents, tasks, action, mask = (torch.stack(jobResults[k]) for k in range(4))
    reward, agent_id, adv = torch.stack(jobResults[4]), torch.stack(jobResults[5]), torch.stack(jobResults[6])
    N = agents.shape[0]
    assert agents.shape == (N, A, 6) and tasks.shape == (N, T + 1, 5) and action.shape == (N, 1) and action.dtype == torch.int64
    assert mask.shape == (N, T + 1) and mask.dtype == torch.bool and reward.shape == (N, 1)
    assert agent_id.shape == (N, 1, 1) and agent_id.dtype == torch.int64 and adv.shape == (N, 1)
    logp = r.localNetwork(tasks, agents, mask)
    assert torch.gather(logp, 1, action).shape == (N, 1)                        # driver.py:176
    assert not mask.gather(1, action).any()                                      # sampled actions are never masked
    summary, greedy = r.last["summary"].cpu().numpy(), r.last["greedy_summary"].cpu().numpy()
    nz = reward[:, 0].nonzero()[:, 0].cpu().numpy()
    assert len(nz) == B and np.allclose(reward[nz, 0].cpu().numpy(), summary[:, 0].astype(np.float32))  # worker.py:91
    np.testing.assert_allclose(metrics["makespan"], summary[:, 3].mean())
    # replay every episode through the oracle with the recorded actions (leaders/followers from the shared protocol)
    first = 2 * B
    inst = generate_batch(B, A, T, base_seed=11, first=first)
    seeds = env_seeds(11, first, B)
    ends = np.concatenate([[-1], nz])
    ag, tk, mk, ac, ld = (x.cpu().numpy() for x in (agents, tasks, mask, action[:, 0], agent_id[:, 0, 0]))
    for b in range(B):
        lo, hi = ends[b] + 1, ends[b + 1] + 1
        o = oracle_lib.OracleEnv(A, T).load(inst["depot"][b], inst["task_xy"][b], inst["req"][b], inst["dur"][b])
        ref = o.rollout(int(seeds[b]), 0, oracle_lib.POLICY_INJECTED, cap_steps=4096, inj_action=ac[lo:hi].astype(np.int32))
        assert ref["n_steps"] == hi - lo
        assert np.array_equal(ref["leader"], ld[lo:hi]) and np.array_equal(ref["agents_obs"], ag[lo:hi])
        assert np.array_equal(ref["tasks_obs"], tk[lo:hi]) and np.array_equal(ref["mask"], mk[lo:hi].astype(np.uint8))
        assert ref["reward"] == summary[b, 0]
        a = adv[lo:hi, 0].cpu().numpy()
        assert np.allclose(a, np.float32(summary[b, 0] - greedy[b, 0]))          # worker.py:92-101


def test_testing_greedy_is_deterministic(gpu_device):
    from dcmrta_amd.policy import AttentionNet
    from dcmrta_amd.runner import BatchedRunner
    torch.manual_seed(1)
    r = BatchedRunner(n_envs=4, device=gpu_device, net_factory=lambda: AttentionNet(6, 5, 32))
    a = r.testing(8, 12, seeds=range(6))
    b = r.testing(8, 12, seeds=range(6))
    assert a.shape == (6,) and np.array_equal(a, b) and (a < 0).all()
    assert isinstance(r.testing(8, 12, seed=3), float) and r.testing(8, 12, seed=3) == a[3]


def test_discounted_advantage(gpu_device):
    """discount(x, gamma) of worker.py:14-15 for gamma != 1: reverse discounted cumsum of [0,...,0,adv]."""
    import scipy.signal as signal
    from dcmrta_amd.policy import AttentionNet
    from dcmrta_amd.runner import BatchedRunner
    torch.manual_seed(2)
    r = BatchedRunner(n_envs=3, device=gpu_device, net_factory=lambda: AttentionNet(6, 5, 32), gamma=0.9)
    w = r.get_weights()
    res, _, _ = r.job(w, w, 0, 6, 9)
    adv, rew = res[6][:, 0].cpu().numpy(), res[4][:, 0].cpu().numpy()
    ends = np.flatnonzero(rew != 0)
    lo = 0
    for b, hi in enumerate(ends):
        x = np.zeros(hi - lo + 1, np.float32)
        x[-1] = adv[hi]
        ref = signal.lfilter([1], [1, -0.9], x[::-1], axis=0)[::-1]
        np.testing.assert_allclose(adv[lo:hi + 1], ref, rtol=1e-5)
        lo = hi + 1


def test_ragged_job_and_testing(gpu_device, oracle_lib):
    """(lo, hi) ranges: every env of the job draws its own sizes like TaskEnv(agents_range, tasks_range, seed)
    (env/task_env.py:58-65); the padded experience replays through the oracle at each env's own size."""
    from dcmrta_amd.choice import env_seeds
    from dcmrta_amd.instances import generate_batch_ranges
    from dcmrta_amd.policy import AttentionNet
    from dcmrta_amd.runner import BatchedRunner
    torch.manual_seed(4)
    B, AR, TR = 8, (4, 9), (6, 15)
    r = BatchedRunner(n_envs=B, device=gpu_device, net_factory=lambda: AttentionNet(6, 5, 32), base_seed=21)
    w = {k: v.clone() for k, v in r.get_weights().items()}
    res, metrics, _ = r.job(w, w, 1, AR, TR)
    agents, tasks, action, mask, reward, agent_id = (res[k].cpu().numpy() for k in range(6))
    assert agents.shape[1:] == (AR[1], 6) and tasks.shape[1:] == (TR[1] + 1, 5) and mask.shape[1] == TR[1] + 1
    inst = generate_batch_ranges(range(21 + B, 21 + 2 * B), AR, TR)
    assert len(set(inst["n_agents"].tolist())) > 1 and len(set(inst["n_tasks"].tolist())) > 1
    seeds = env_seeds(21, B, B)
    summary = r.last["summary"].cpu().numpy()
    ends = np.concatenate([[-1], np.flatnonzero(reward[:, 0] != 0)])
    assert len(ends) == B + 1
    for b in range(B):
        a, t = int(inst["n_agents"][b]), int(inst["n_tasks"][b])
        lo, hi = ends[b] + 1, ends[b + 1] + 1
        o = oracle_lib.OracleEnv(a, t).load(inst["depot"][b], inst["task_xy"][b, :t], inst["req"][b, :t], inst["dur"][b, :t])
        ref = o.rollout(int(seeds[b]), 0, oracle_lib.POLICY_INJECTED, cap_steps=4096, inj_action=action[lo:hi, 0].astype(np.int32))
        assert ref["n_steps"] == hi - lo and ref["reward"] == summary[b, 0]
        assert np.array_equal(ref["leader"], agent_id[lo:hi, 0, 0])
        assert np.array_equal(ref["agents_obs"], agents[lo:hi, :a]) and np.all(agents[lo:hi, a:] == -1)
        assert np.array_equal(ref["tasks_obs"], tasks[lo:hi, :t + 1]) and np.all(tasks[lo:hi, t + 1:] == -1)
        assert np.array_equal(ref["mask"], mask[lo:hi, :t + 1].astype(np.uint8)) and mask[lo:hi, t + 1:].all()
        assert (action[lo:hi, 0] <= t).all()
    # testing() with ranges (runner.py:45-49 defaults): ragged greedy evaluation, deterministic, one reward per seed
    rewards = r.testing(AR, TR, seeds=range(40, 46))
    assert rewards.shape == (6,) and (rewards < 0).all()
    assert np.array_equal(rewards, r.testing(AR, TR, seeds=range(40, 46)))
    assert r.testing(AR, TR, seed=43) == rewards[3]


def test_run_test_and_run_test_is(gpu_device, oracle_lib, golden_dir):
    """BatchedRunner.run_test: Worker.run_test (worker.py:114-157) and run_test_IS (:159-198) over the shipped test-set
    instances in one batch.  The individual-selection run needs no injected choices (the device takes the lowest pending id,
    no followers) and equals the reference loop restated on the oracle with the same greedy policy decisions."""
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.instances import load_instances_npz
    from dcmrta_amd.policy import AttentionNet
    from dcmrta_amd.runner import METRIC_KEYS, BatchedRunner
    torch.manual_seed(7)
    inst, A = load_instances_npz(os.path.join(golden_dir, "instances_20A50T.npz"))
    sub = {k: v[:6] for k, v in inst.items()}
    r = BatchedRunner(n_envs=6, device=gpu_device, net_factory=lambda: AttentionNet(6, 5, 32))
    for is_mode in (False, True):
        m = r.run_test(sub, n_agents=A, individual_selection=is_mode)
        assert set(m) == set(METRIC_KEYS) | {"reward"} and all(v.shape == (6,) for v in m.values())
        assert np.array_equal(m["reward"], -m["makespan"]) and (m["success_rate"] <= 1).all()
        again = r.run_test(sub, n_agents=A, individual_selection=is_mode)
        assert all(np.array_equal(m[k], again[k], equal_nan=True) for k in m)
    # individual selection, stepwise against the oracle: same decisions -> same final state
    env = BatchedTaskEnv(1, A, 50, device=gpu_device, individual_selection=True)
    env.load_instances(**{k: v[:1] for k, v in sub.items()})
    obs = env.reset(np.array([5], np.uint64))
    o = oracle_lib.OracleEnv(A, 50).load(sub["depot"][0], sub["task_xy"][0], sub["req"][0], sub["dur"][0])
    n, finished = 0, False
    while not finished and o.now < 100:                                       # worker.py:163
        ids, t = o.next_decision()
        o.now = t
        o.task_update(); o.agent_update()
        for a in ids:                                                        # :170 ascending ids, each alone
            assert bool(obs.active[0]) and int(obs.leader[0]) == int(a)      # the device offers the same agent
            assert np.array_equal(obs.mask[0, 1:].cpu().numpy().astype(np.uint8), o.mask()[1:])
            action = int(torch.argmax((~obs.mask[0]).to(torch.int32)))       # first valid action
            o.agent_step(int(a), action)
            o.task_update(); o.agent_update()
            obs = env.step(torch.tensor([action], dtype=torch.int32))
            n += 1
        finished = o.check_finished()
    oracle_lib.lib().orc_finish_episode(o._h)
    assert not bool(obs.active[0]) and n > 20
    assert env.summary()[0, 0].item() == o.final()["reward"]
