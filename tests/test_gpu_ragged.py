"""-m gpu: ragged batches (dcm_load_instances_ragged) -- every env of a batch has its own (A_e, T_e), as
TaskEnv(agents_range=(10,20), tasks_range=(20,50), seed=s) draws them (env/task_env.py:58-65; Runner.testing,
runner.py:45-49).  Each env must equal the oracle run at its own sizes, bit for bit, and the rows beyond its sizes
must be padding in the policy's convention (attention.py:10-18, worker.py:253-261)."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu

TASK_KEYS = ("finished", "feasible", "time_start", "time_finish", "task_wait", "n_members", "n_abandoned")
AGENT_KEYS = ("travel_dist", "returned", "agent_wait")


def _ragged(seeds, ar=(10, 20), tr=(20, 50)):
    from dcmrta_amd.instances import generate_batch_ranges
    return generate_batch_ranges(seeds, ar, tr)


def _oracle(inst, b, mwt=10.0):
    import oracle
    a, t = int(inst["n_agents"][b]), int(inst["n_tasks"][b])
    return oracle.OracleEnv(a, t, max_waiting_time=mwt).load(inst["depot"][b], inst["task_xy"][b, :t], inst["req"][b, :t],
                                                             inst["dur"][b, :t])


def _slice_final(fin, a, t):
    out = dict(fin)
    for k in TASK_KEYS:
        out[k] = fin[k][:t]
    for k in AGENT_KEYS:
        out[k] = fin[k][:a]
    return out


def _custom(sizes, A, T, seed=5):
    """Hand-picked sizes (incl. 1 agent / 1 task / the batch maximum) with random instances."""
    rng = np.random.default_rng(seed)
    B = len(sizes)
    inst = dict(depot=rng.random((B, 2)), task_xy=rng.random((B, T, 2)), req=rng.integers(1, 6, (B, T)).astype(np.int32),
                dur=np.full((B, T), 5.0), n_agents=np.array([s[0] for s in sizes], np.int32),
                n_tasks=np.array([s[1] for s in sizes], np.int32))
    return inst


def test_ragged_rollout_matches_oracle_per_env(gpu_device):
    import oracle
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.choice import env_seeds
    B = 48
    inst = _ragged(range(100, 100 + B))
    assert len(set(zip(inst["n_agents"].tolist(), inst["n_tasks"].tolist()))) > 20      # really ragged
    seeds = env_seeds(9, 0, B)
    env = BatchedTaskEnv(B, 20, 50, device=gpu_device).load_instances(**inst)
    env.reset(seeds, observe=False)
    steps = env.rollout_random(1).cpu().numpy()
    fin = H.gpu_final(env)
    for b in range(B):
        ref = _oracle(inst, b).rollout(int(seeds[b]), 0, oracle.POLICY_RANDOM, cap_steps=100000, record=False)
        assert steps[b] == ref["n_steps"], b
        a, t = int(inst["n_agents"][b]), int(inst["n_tasks"][b])
        H.assert_final_matches(_slice_final(fin[b], a, t), ref, f"env {b} ({a}A/{t}T)")
        for k in TASK_KEYS:
            assert not np.any(fin[b][k][t:]), (b, k)                                    # getter rows beyond T_e read 0
        for k in AGENT_KEYS:
            assert not np.any(fin[b][k][a:]), (b, k)


@pytest.mark.parametrize("shape", [(20, 50), (7, 70), (128, 130)])
def test_ragged_extreme_sizes(gpu_device, shape):
    """1 agent, 1 task, the batch maximum and sizes straddling the 64-lane chunks inside one batch."""
    import oracle
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.choice import env_seeds
    A, T = shape
    sizes = [(1, 1), (A, T), (1, T), (A, 1), (min(A, 3), min(T, 2)), (A, max(1, T - 1)), (max(1, A - 1), T),
             (min(A, 64), min(T, 64)), (min(A, 65), min(T, 65)), (max(1, A // 2), max(1, T // 2))]
    inst = _custom(sizes, A, T)
    B = len(sizes)
    seeds = env_seeds(31, 0, B)
    env = BatchedTaskEnv(B, A, T, device=gpu_device).load_instances(**inst)
    env.reset(seeds, observe=False)
    steps = env.rollout_random(2).cpu().numpy()              # two episodes: the restart path as well
    fin = H.gpu_final(env)
    for b, (a, t) in enumerate(sizes):
        o = _oracle(inst, b)
        r1 = o.rollout(int(seeds[b]), 0, oracle.POLICY_RANDOM, cap_steps=100000, record=False)
        o.clear_decisions()
        r2 = o.rollout(int(seeds[b]), r1["n_steps"], oracle.POLICY_RANDOM, cap_steps=100000, record=False)
        assert steps[b] == r1["n_steps"] + r2["n_steps"], (b, a, t)
        H.assert_final_matches(_slice_final(fin[b], a, t), r2, f"env {b} ({a}A/{t}T)")


def test_ragged_lockstep_observations_and_padding(gpu_device):
    """The lockstep API on a ragged batch: every observation row / mask entry inside an env's sizes equals the oracle's,
    everything beyond them is padding (-1 rows, mask True), at every decision."""
    import oracle
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.choice import env_seeds
    B = 24
    inst = _ragged(range(7, 7 + B))
    seeds = env_seeds(4, 0, B)
    env = BatchedTaskEnv(B, 20, 50, device=gpu_device).load_instances(**inst)
    got = H.run_lockstep(env, seeds, lambda b, i, m, l: H.host_random_action(m, int(seeds[b]), i))
    fin = H.gpu_final(env)
    for b in range(B):
        a, t = int(inst["n_agents"][b]), int(inst["n_tasks"][b])
        ref = _oracle(inst, b).rollout(int(seeds[b]), 0, oracle.POLICY_RANDOM, cap_steps=100000, record=True)
        g = got[b]
        assert g["n_steps"] == ref["n_steps"], b
        assert np.array_equal(g["leader"], ref["leader"]) and np.array_equal(g["action"], ref["action"]), b
        assert np.array_equal(g["now"], ref["now"]), b
        assert np.array_equal(g["mask"][:, :t + 1], ref["mask"]), b
        assert np.array_equal(g["agents_obs"][:, :a], ref["agents_obs"]), b
        assert np.array_equal(g["tasks_obs"][:, :t + 1], ref["tasks_obs"]), b
        assert np.all(g["mask"][:, t + 1:] == 1), b
        assert np.all(g["agents_obs"][:, a:] == -1.0) and np.all(g["tasks_obs"][:, t + 1:] == -1.0), b
        H.assert_final_matches(_slice_final(fin[b], a, t), ref, f"env {b}")
    # finished envs keep the padding in their (inactive) observation rows
    obs = env.observe()
    ag, mk = obs.agents.cpu().numpy(), obs.mask.cpu().numpy()
    for b in range(B):
        a, t = int(inst["n_agents"][b]), int(inst["n_tasks"][b])
        assert np.all(ag[b, a:] == -1.0) and np.all(mk[b, t + 1:] == 1)


def test_ragged_with_full_sizes_equals_uniform(gpu_device):
    """sizes == the batch (A,T) everywhere: the ragged code path (runtime-shape kernels) gives the uniform batch's result."""
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.choice import env_seeds
    from dcmrta_amd.instances import generate_batch
    B, A, T = 16, 20, 50
    inst = generate_batch(B, A, T, base_seed=12)
    seeds = env_seeds(2, 0, B)
    outs = []
    for ragged in (False, True):
        kw = dict(n_agents=np.full(B, A, np.int32), n_tasks=np.full(B, T, np.int32)) if ragged else {}
        env = BatchedTaskEnv(B, A, T, device=gpu_device).load_instances(**inst, **kw)
        env.reset(seeds, observe=False)
        steps = env.rollout_random(3).cpu().numpy()
        outs.append((steps, env.summary().cpu().numpy()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1], equal_nan=True)
    # and loading a uniform batch again leaves ragged mode
    env.load_instances(**inst)
    env.reset(seeds, observe=False)
    assert np.array_equal(env.rollout_random(3).cpu().numpy(), outs[0][0])


def test_ragged_argument_errors(gpu_device):
    from dcmrta_amd.batched_env import BatchedTaskEnv, DcmError
    B, A, T = 4, 6, 9
    inst = _custom([(1, 1), (6, 9), (2, 3), (3, 2)], A, T)
    env = BatchedTaskEnv(B, A, T, device=gpu_device)
    for bad in (dict(n_agents=[0, 6, 2, 3]), dict(n_agents=[7, 6, 2, 3]), dict(n_tasks=[1, 10, 3, 2]), dict(n_tasks=[1, 9, 0, 2])):
        kw = dict(inst)
        kw.update({k: np.array(v, np.int32) for k, v in bad.items()})
        with pytest.raises(DcmError):
            env.load_instances(**kw)
    with pytest.raises(DcmError):
        env.load_instances(inst["depot"], inst["task_xy"], inst["req"], inst["dur"], n_agents=inst["n_agents"])
    env.load_instances(**inst)
    env.load_routes([[[0] for _ in range(A)] for _ in range(B)])
    with pytest.raises(DcmError):
        env.execute_routes()


@pytest.mark.parametrize("mwt", [10.0, 1.0, 0.25])
def test_terminal_metrics_lists_and_their_fallbacks(gpu_device, mwt):
    """The per-agent waiting-time sums (env/task_env.py:358-364) in the 20A/50T layout, whose kernels gather every agent's member
    terms into a list of at most 14 entries and count at most eight of its abandonment entries per task: few agents x many tasks
    (lists longer than 14), a short max_waiting_time (more than 8, and more than the log's 16, abandonments per agent) and the
    plain shape in one batch -- persistent kernel (two episodes) and lockstep API, every env bit-equal to the oracle."""
    import oracle
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.choice import env_seeds
    sizes = [(20, 50), (2, 50), (3, 50), (1, 50), (4, 45), (20, 50), (6, 50), (2, 17), (12, 50), (5, 50), (20, 12), (8, 50)]
    inst = _custom(sizes, 20, 50, seed=11)
    B = len(sizes)
    for b, (a, t) in enumerate(sizes):                        # few agents: requirements they can meet, so that tasks finish and list them
        if a <= 3:
            inst["req"][b, :] = 1
        elif a <= 6:
            inst["req"][b, :] = np.minimum(inst["req"][b, :], 2)
    seeds = env_seeds(77, 0, B)
    env = BatchedTaskEnv(B, 20, 50, device=gpu_device, max_waiting_time=mwt).load_instances(**inst)
    env.reset(seeds, observe=False)
    steps = env.rollout_random(2).cpu().numpy()
    fin = H.gpu_final(env)
    long_lists = many = 0
    for b, (a, t) in enumerate(sizes):
        o = _oracle(inst, b, mwt)
        r1 = o.rollout(int(seeds[b]), 0, oracle.POLICY_RANDOM, cap_steps=100000, record=False)
        o.clear_decisions()
        r2 = o.rollout(int(seeds[b]), r1["n_steps"], oracle.POLICY_RANDOM, cap_steps=100000, record=False)
        assert steps[b] == r1["n_steps"] + r2["n_steps"], (b, a, t)
        H.assert_final_matches(_slice_final(fin[b], a, t), r2, f"env {b} ({a}A/{t}T, mwt {mwt})")
        long_lists += int(np.sum(r2["n_members"]) > 14 * a)
        many += int(np.sum(r2["n_abandoned"]) > 8 * a)
    assert long_lists >= 1                                    # some agent is listed by more than 14 tasks (pigeonhole)
    if mwt < 1.0:
        assert many >= 1                                      # some agent was abandoned more than eight times (pigeonhole)
    # the same batch through the lockstep kernel (k_step_fast), one episode
    env2 = BatchedTaskEnv(B, 20, 50, device=gpu_device, max_waiting_time=mwt).load_instances(**inst)
    H.run_lockstep(env2, seeds, lambda b, i, m, l: H.host_random_action(m, int(seeds[b]), i))
    fin2 = H.gpu_final(env2)
    for b, (a, t) in enumerate(sizes):
        ref = _oracle(inst, b, mwt).rollout(int(seeds[b]), 0, oracle.POLICY_RANDOM, cap_steps=100000, record=False)
        H.assert_final_matches(_slice_final(fin2[b], a, t), ref, f"lockstep env {b} ({a}A/{t}T, mwt {mwt})")
