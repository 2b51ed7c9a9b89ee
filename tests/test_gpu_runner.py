"""-m gpu: BatchedRunner.job keeps the reference runner's contract (runner.py:58-71, driver.py:135-176) and its
experience replays bit-exactly through the oracle (same choices -> same observations)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("twin", [False, True])
def test_job_contract_and_replay(gpu_device, oracle_lib, twin):
    """twin=True: the sampled episodes and their greedy twins played as ONE batch of 2 B envs (BatchedRunner(twin_rollout=True))."""
    from dcmrta_amd.choice import env_seeds
    from dcmrta_amd.instances import generate_batch
    from dcmrta_amd.policy import AttentionNet
    from dcmrta_amd.runner import METRIC_KEYS, BatchedRunner
    torch.manual_seed(0)
    B, A, T = 6, 10, 20
    small = lambda: AttentionNet(6, 5, 32)
    r = BatchedRunner(metaAgentID=3, n_envs=B, device=gpu_device, net_factory=small, base_seed=11, twin_rollout=twin)
    r.keep_greedy_record = True
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
    # the greedy self-critic twin (worker.py:89,200-235: argmax of the same net on the same instance) is an episode of the
    # reference env too: replay its recorded decisions through the oracle
    _replay_recorded(oracle_lib, r.last["greedy_rec"], greedy, inst, seeds, A, T)
    # ... and the argmax really was the greedy choice of the net on the recorded observations
    g = r.last["greedy_rec"]
    act = g["active"]
    lp = r.localNetwork(g["tasks"][act], g["agents"][act], g["mask"][act])
    assert (lp.gather(1, g["action"][act].unsqueeze(1))[:, 0] >= lp.max(1).values - 1e-4).all()


def _replay_recorded(oracle_lib, rec, summary, inst, seeds, A, T, n_agents=None, n_tasks=None):
    """Every episode of a recorded batched rollout ([S, B, ...] experience + `active`) through the oracle with the recorded
    actions injected: leaders, observations, masks, reward and metrics must be the reference env's."""
    act = rec["active"].cpu().numpy()
    ag, tk, mk, ac, ld = (rec[k].cpu().numpy() for k in ("agents", "tasks", "mask", "action", "leader"))
    summary = np.asarray(summary)
    for b in range(act.shape[1]):
        a = int(n_agents[b]) if n_agents is not None else A
        t = int(n_tasks[b]) if n_tasks is not None else T
        sel = act[:, b]
        o = oracle_lib.OracleEnv(a, t).load(inst["depot"][b], inst["task_xy"][b, :t], inst["req"][b, :t], inst["dur"][b, :t])
        ref = o.rollout(int(seeds[b]), 0, oracle_lib.POLICY_INJECTED, cap_steps=8192, inj_action=ac[sel, b].astype(np.int32))
        assert ref["n_steps"] == int(sel.sum()), b
        assert np.array_equal(ref["leader"], ld[sel, b]), b
        assert np.array_equal(ref["agents_obs"], ag[sel, b, :a]) and np.array_equal(ref["tasks_obs"], tk[sel, b, :t + 1]), b
        assert np.array_equal(ref["mask"], mk[sel, b, :t + 1].astype(np.uint8)), b
        assert ref["reward"] == summary[b, 0], b
        for i in range(6):
            assert ref["metrics"][i] == summary[b, 2 + i], (b, i)


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
    r.keep_greedy_record = True
    from dcmrta_amd.choice import env_seeds
    seeds = env_seeds(r.base_seed, 0, 6)
    for is_mode in (False, True):
        m = r.run_test(sub, n_agents=A, individual_selection=is_mode)
        assert set(m) == set(METRIC_KEYS) | {"reward"} and all(v.shape == (6,) for v in m.values())
        assert np.array_equal(m["reward"], -m["makespan"]) and (m["success_rate"] <= 1).all()
        rec, sm = r.last["rec"], r.last["summary"].cpu().numpy()
        for k, name in enumerate(METRIC_KEYS):
            assert np.array_equal(m[name], sm[:, 2 + k])
        if not is_mode:
            # leader-follower mode (worker.py:114-157): the recorded greedy episodes are episodes of the reference env
            _replay_recorded(oracle_lib, rec, sm, sub, seeds, A, 50)
        else:
            # individual selection (worker.py:159-198): the reference loop restated on the oracle with the recorded actions
            act, ac, ld = (rec[k].cpu().numpy() for k in ("active", "action", "leader"))
            for b in range(6):
                o = oracle_lib.OracleEnv(A, 50).load(sub["depot"][b], sub["task_xy"][b], sub["req"][b], sub["dur"][b])
                acts, leads = ac[act[:, b], b], ld[act[:, b], b]
                n, finished = 0, False
                while not finished and o.now < 100:                              # worker.py:163
                    ids, t = o.next_decision()
                    o.now = t
                    o.task_update(); o.agent_update()
                    for a in ids:                                                # :170 ascending ids, each alone
                        assert leads[n] == a, (b, n)
                        o.agent_step(int(a), int(acts[n]))
                        o.task_update(); o.agent_update()
                        n += 1
                    finished = o.check_finished()
                oracle_lib.lib().orc_finish_episode(o._h)
                f = o.final()
                assert n == len(acts) and f["reward"] == sm[b, 0], b
                for i in range(6):
                    assert f["metrics"][i] == sm[b, 2 + i], (b, i)
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


def test_masked_action_policy_is_reported(gpu_device):
    """A policy that ignores the mask makes the device freeze envs (DCM_FLAG_BAD_ACTION) whose summary rows stay NaN: the
    runner must raise instead of averaging NaN rewards into the batch."""
    from dcmrta_amd.runner import BatchedRunner, EnvError

    class IgnoresMask(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(1))

        def forward(self, tasks, agents, mask):           # always prefers task 1, masked or not
            lp = torch.full(mask.shape, -5.0, device=mask.device) + self.w
            lp[:, 1] = 0.0
            return torch.log_softmax(lp, dim=1)
    r = BatchedRunner(n_envs=4, device=gpu_device, net_factory=IgnoresMask)
    w = r.get_weights()
    with pytest.raises(EnvError, match="frozen by the device"):
        r.job(w, w, 0, 6, 9)


def test_actor_results_are_consumable_by_a_learner(gpu_device):
    """Contract of what the actor shell hands to a learner (the consumer is driver.py, which is not part of this
    repo): every job returns nine per-decision lists of equal length that concatenate across jobs and stack into batch tensors
    of the policy's input shapes (seven of the nine slots are in use); the action slot indexes the policy's log-probabilities; a gradient step on a REINFORCE-style
    objective built from those tensors is finite and moves the weights; evaluation actors answer testing(seed=...) with one float."""
    from dcmrta_amd import ray_compat as ray
    from dcmrta_amd.policy import AttentionNet
    from dcmrta_amd.ray_compat import RLRunner
    torch.manual_seed(3)
    n_actors, batch, A, T = 2, 1024, 12, 23
    make_net = lambda: AttentionNet(6, 5, 32)
    ray.init(n_envs=24, net_factory=make_net, base_seed=5)
    dev = torch.device(gpu_device)
    learner, baseline = make_net().to(dev), make_net().to(dev)
    opt = torch.optim.Adam(learner.parameters(), lr=1e-4)
    start = {k: v.clone() for k, v in learner.state_dict().items()}
    actors = [RLRunner.remote(i) for i in range(n_actors)]
    pool = [[] for _ in range(9)]          # slot order: agent obs, task obs, action, mask, reward, leader index, advantage, ... (worker.py:77-83)
    episode, n_updates, makespans = 0, 0, []
    for _round in range(2):
        handles = []
        for actor in actors:
            handles.append(actor.job.remote(learner.state_dict(), baseline.state_dict(), episode, A, T))
            episode += 1
        ready, pending = ray.wait(handles, num_returns=n_actors)
        assert not pending
        for slots, metrics, info in ray.get(ready):
            # one entry per recorded decision in the seven slots in use; the last two stay empty (worker.py:20-21 keeps nine)
            assert len(slots) == 9 and len({len(x) for x in slots[:7]}) == 1 and len(slots[0]) > 0 and slots[7] == [] and slots[8] == []
            assert {"id", "episode_number"} <= set(info)
            assert set(metrics) == {"success_rate", "makespan", "time_cost", "waiting_time", "travel_dist", "efficiency"}
            makespans.append(metrics["makespan"])
            for k in range(9):
                pool[k].extend(slots[k])                                                           # plain list concatenation works
        while len(pool[0]) >= batch:
            take = [x[:batch] for x in pool]
            pool = [x[batch:] for x in pool]
            agents_in, tasks_in = torch.stack(take[0]), torch.stack(take[1])
            action, mask = torch.stack(take[2]), torch.stack(take[3])
            reward, leader, advantage = torch.stack(take[4]), torch.stack(take[5]), torch.stack(take[6])
            assert agents_in.shape == (batch, A, 6) and agents_in.dtype == torch.float32
            assert tasks_in.shape == (batch, T + 1, 5) and tasks_in.dtype == torch.float32
            assert mask.shape == (batch, T + 1) and mask.dtype == torch.bool
            assert action.shape == (batch, 1) and action.dtype == torch.int64 and int(action.min()) >= 0 and int(action.max()) <= T
            assert leader.shape == (batch, 1, 1) and reward.shape == (batch, 1) and advantage.shape[0] == batch
            logp_all = learner(tasks_in, agents_in, mask)
            chosen = logp_all.gather(1, action)                       # the action slot addresses the policy output directly
            assert bool((chosen > -9000).all())                       # no recorded action was a masked one
            loss = -(chosen * advantage.detach()).mean()
            opt.zero_grad()
            loss.backward()
            gnorm = torch.nn.utils.clip_grad_norm_(learner.parameters(), 10.0)
            opt.step()
            assert torch.isfinite(loss) and torch.isfinite(gnorm) and float(gnorm) > 0
            assert all(q.grad is not None and bool(torch.isfinite(q.grad).all()) for q in learner.parameters())
            n_updates += 1
    assert n_updates >= 2 and np.isfinite(np.nanmean(makespans))
    end = learner.state_dict()
    assert any(not torch.equal(start[k], end[k]) for k in start)      # the learner really moved
    for actor in actors:
        ray.kill(actor)
    # evaluation side: fresh actors take baseline weights and answer a seeded test episode with its return
    testers = [RLRunner.remote(metaAgentID=i) for i in range(n_actors)]
    for t in testers:
        ray.get(t.set_baseline_weights.remote(baseline.state_dict()))
    answers = ray.get(ray.wait([t.testing.remote(seed=1000 + j) for j, t in enumerate(testers)], num_returns=n_actors)[0])
    assert len(answers) == n_actors and all(isinstance(x, float) and x < 0 for x in answers)
    for t in testers:
        ray.kill(t)
    # a call that fails inside the actor surfaces from get
    broken = RLRunner.remote(0)
    with pytest.raises(Exception):
        ray.get(broken.job.remote({}, {}, 0, 5, 8))
    ray.kill(broken)


def test_rl_test_example_script(gpu_device, tmp_path):
    """examples/rl_test.py = RL_test.py:1-51 batched: both METHODs over the shipped test-set instances, CSV in the reference's
    column layout."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for method in ("LF", "IA"):
        out = tmp_path / f"REINFORCE_{method}.csv"
        r = subprocess.run([sys.executable, os.path.join(root, "examples", "rl_test.py"), "--method", method, "--out", str(out)],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        rows = out.read_text().strip().splitlines()
        assert rows[0] == ",success_rate,makespan,time_cost,waiting_time,travel_dist,efficiency" and len(rows) == 51
        vals = np.array([[float(x) for x in l.split(",")[1:]] for l in rows[1:]])
        assert (vals[:, 0] >= 0).all() and (vals[:, 0] <= 1).all() and (vals[:, 1] > 0).all()


def test_concurrent_actors_capture_while_others_replay(gpu_device):
    """ray_compat.init(concurrent=True): one worker thread per actor, here two actors on the one GPU with REAL graph-captured
    jobs.  Training draws a new batch shape every round (driver.py:114-115), so both threads create env handles (hipMalloc)
    and capture HIP graphs while the other one allocates, launches and replays -- the failure shape of a process-wide capture
    mode.  Captures are serialised by graph_rollout.CAPTURE_LOCK and run in capture_error_mode="thread_local"; dcm_create's
    per-device LDS attribute table is mutex-guarded.  The deterministic (greedy) results must equal a single-threaded runner's;
    unequal shards of one env budget must not overlap (common episode stride)."""
    from dcmrta_amd import ray_compat as ray
    from dcmrta_amd.policy import AttentionNet
    from dcmrta_amd.runner import BatchedRunner
    torch.manual_seed(5)
    ray.init(total_envs=45, num_actors=2, devices=[torch.device(gpu_device)], concurrent=True, base_seed=21,
             net_factory=lambda: AttentionNet(6, 5, 32))
    actors = [ray.RLRunner.remote(i) for i in range(2)]
    sizes = [a._obj._r.B for a in actors]
    assert sizes == [23, 22] and all(a._obj._r.episode_stride == 23 for a in actors)
    w = {k: v.clone() for k, v in ray.get(actors[0].get_weights.remote()).items()}
    ep = 0
    for A, T in ((8, 12), (10, 20), (6, 9), (8, 12)):                    # three new shapes (captures), one cached shape
        jobs = []
        for a in actors:
            jobs.append(a.job.remote(w, w, ep, A, T))
            ep += 1
        done, rest = ray.wait(jobs, num_returns=2)
        assert rest == []
        for (res, metrics, info), n in zip(ray.get(jobs), sizes):
            agents = torch.stack(res[0])
            assert agents.shape[1:] == (A, 6) and agents.shape[0] >= n and np.isfinite(metrics["makespan"])
            assert int((torch.stack(res[4])[:, 0] != 0).sum()) == n          # one terminal reward per episode of the shard
    # instance blocks: job e of actor i starts at e * 23 -- disjoint although actor 1 only holds 22 envs
    assert actors[0]._obj._r.first_env(2) == 46 and actors[1]._obj._r.first_env(1) == 23
    # greedy evaluation (deterministic) on both threads at once == a plain single-threaded runner with the same weights
    seeds = list(range(40, 52))
    refs = [a.testing.remote(seed=s) for s in seeds[:6] for a in actors[:1]] + [actors[1].testing.remote(seed=s) for s in seeds[6:]]
    got = ray.get(refs)
    ray.shutdown()
    single = BatchedRunner(n_envs=4, device=gpu_device, net_factory=lambda: AttentionNet(6, 5, 32), base_seed=21)
    single.set_weights(w)
    exp = single.testing(seeds=seeds)
    single.close()
    assert np.array_equal(np.array(got), exp)


def test_tuned_gemms_job_replays_through_the_oracle(gpu_device, oracle_lib, tmp_path):
    """BatchedRunner(tune_gemms=True): the policy is run with TunableOp tuning on before the rollout graphs are captured, the graphs
    then use the selected GEMMs; the recorded episodes are still episodes of the reference env."""
    import torch.cuda.tunable as tunable
    from dcmrta_amd.choice import env_seeds
    from dcmrta_amd.instances import generate_batch
    from dcmrta_amd.policy import AttentionNet
    from dcmrta_amd.runner import BatchedRunner
    torch.manual_seed(1)
    B, A, T = 8, 10, 20
    r = BatchedRunner(n_envs=B, device=gpu_device, net_factory=lambda: AttentionNet(6, 5, 32), base_seed=3, tune_gemms=True,
                      gemm_tuning_file=str(tmp_path / "tunableop.csv"))
    try:
        assert tunable.is_enabled() and not tunable.tuning_is_enabled()
        r.keep_greedy_record = True
        w = {k: v.clone() for k, v in r.get_weights().items()}
        r.job(w, w, episodeNumber=0, agents_num=A, tasks_num=T, as_lists=False)
        assert not tunable.tuning_is_enabled()                       # tuning only around the warm-up forwards
        inst = generate_batch(B, A, T, base_seed=3, first=0)
        _replay_recorded(oracle_lib, r.last["greedy_rec"], r.last["greedy_summary"].cpu().numpy(), inst, env_seeds(3, 0, B), A, T)
    finally:
        r.close()
        tunable.enable(False)
