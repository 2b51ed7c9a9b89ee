"""-m gpu: the persistent rollout kernel (dcm_rollout_random, the kernel the headline number is quoted on) against the
oracle AT EVERY KIND OF OUTPUT: not only step counts and terminal arrays but the per-decision observation tensors and
mask it stores (worker.py:57-68), checked at arbitrary decision indices through the decision budget of the ABI."""
import json
import os

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def _oracle_traces(oracle_lib, inst, seeds, A, T, B, cap=20000):
    out = []
    for b in range(B):
        o = oracle_lib.OracleEnv(A, T).load(inst["depot"][b], inst["task_xy"][b], inst["req"][b], inst["dur"][b])
        out.append(o.rollout(int(seeds[b]), 0, oracle_lib.POLICY_RANDOM, cap_steps=cap, record=True))
    return out


def _check_last_decision(env, refs, taken, tag, ran=None):
    """Every env b has just taken decision taken[b]-1 inside dcm_rollout_random: the observation buffers hold what the
    PERSISTENT KERNEL stored for that decision (its own stores, no other kernel has run since) -- they must be the oracle's
    recorded observation of that decision.  The state it left behind is checked through dcm_observe (a pure function of
    the state): leader, event time and observation of the next pending decision."""
    obs = env.obs()
    ag, tk, mk = (x.clone().cpu().numpy() for x in (obs.agents, obs.tasks, obs.mask))
    st = {k: v.cpu().numpy() for k, v in env.status().items()}
    for b, ref in enumerate(refs):
        k = int(taken[b]) - 1
        if k < 0 or (ran is not None and not ran[b]):   # (no decision in this launch: the buffers are not the kernel's)
            continue
        name = f"{tag} env{b} decision {k}"
        assert np.array_equal(ag[b], ref["agents_obs"][k]), name + ": agents observation"
        assert np.array_equal(tk[b], ref["tasks_obs"][k]), name + ": tasks observation"
        assert np.array_equal(mk[b].astype(np.uint8), ref["mask"][k]), name + ": mask"
        assert st["decisions"][b] == k + 1, name + ": decision counter"
    o2 = env.observe()
    ag2, tk2, mk2, ld2, act2 = (x.cpu().numpy() for x in (o2.agents, o2.tasks, o2.mask, o2.leader, o2.active))
    for b, ref in enumerate(refs):
        k = int(taken[b])
        name = f"{tag} env{b} pending decision {k}"
        if k >= ref["n_steps"]:
            assert not act2[b] and (st["flags"][b] & 1), name + ": episode must be over"
            continue
        assert act2[b] and st["now"][b] == ref["now"][k] and ld2[b] == ref["leader"][k], name + ": time / leader"
        assert np.array_equal(ag2[b], ref["agents_obs"][k]) and np.array_equal(tk2[b], ref["tasks_obs"][k]), name
        assert np.array_equal(mk2[b].astype(np.uint8), ref["mask"][k]), name


@pytest.mark.parametrize("A,T,B,stops", [(20, 50, 48, 6), (50, 200, 32, 4), (15, 35, 32, 5), (12, 23, 16, 4), (33, 60, 12, 3), (70, 130, 6, 3),
                                         (65, 65, 8, 3), (128, 256, 3, 2), (30, 100, 8, 3), (100, 64, 6, 3), (64, 192, 4, 2), (40, 64, 8, 3)])
def test_per_decision_outputs_at_random_indices(gpu_device, oracle_lib, A, T, B, stops):
    """Stop every env after random decision indices (per-env budgets) and compare what the persistent kernel stored for
    that decision with the oracle's recorded rec_agents / rec_tasks / rec_mask (oracle/dcmrta_oracle.c:569-576); carry on;
    finish.  Shapes: BASELINE configs 2 and 4, two training shapes (runtime sizes in the <20,50> layout), and the mid-size class of
    rollout_fast_g.hpp: one / two agent chunks x two / three / four task chunks, T a multiple of 64 (no free lane for the depot)."""
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.choice import env_seeds
    from dcmrta_amd.instances import generate_batch
    inst = generate_batch(B, A, T, base_seed=900 + A)
    seeds = env_seeds(31, 0, B)
    refs = _oracle_traces(oracle_lib, inst, seeds, A, T, B)
    n = np.array([r["n_steps"] for r in refs], np.int64)
    rng = np.random.default_rng(A * 1000 + T)
    env = BatchedTaskEnv(B, A, T, device=gpu_device).load_instances(**inst)
    env.reset(seeds, observe=False)
    taken = np.zeros(B, np.int64)
    for s in range(stops):
        k = taken + rng.integers(0, np.maximum(1, (n - taken + 1) // 2))      # index of the decision to stop after
        if s == 0:
            k[: B // 4] = 0                                                   # the very first decision
            k[B // 4: B // 2] = n[B // 4: B // 2] - 1                         # the very last one (the episode ends on it)
        k = np.minimum(k, n - 1)
        budget = np.maximum(k + 1 - taken, 0)
        steps = env.rollout_random(episodes=1, max_decisions=budget).cpu().numpy()
        assert np.array_equal(steps, budget), (s, steps, budget)
        taken = taken + budget
        _check_last_decision(env, refs, taken, f"{A}A{T}T stop{s}", ran=budget > 0)
    live = taken < n                                                          # finished envs must not start another episode
    steps = env.rollout_random(episodes=1, max_decisions=np.where(live, -1, 0).astype(np.int64)).cpu().numpy()
    assert np.array_equal(steps, n - taken)
    fin = H.gpu_final(env)
    for b in range(B):
        H.assert_final_matches(fin[b], refs[b], f"{A}A{T}T env{b} after budgeted launches")


def test_every_decision_of_an_episode(gpu_device, oracle_lib):
    """Budget 1 per launch: ALL decisions of 8 episodes at 20A/50T, one by one, each store of the persistent kernel
    compared with the oracle's record of that decision."""
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.choice import env_seeds
    from dcmrta_amd.instances import generate_batch
    B, A, T = 8, 20, 50
    inst = generate_batch(B, A, T, base_seed=77)
    seeds = env_seeds(5, 0, B)
    refs = _oracle_traces(oracle_lib, inst, seeds, A, T, B)
    n = np.array([r["n_steps"] for r in refs], np.int64)
    env = BatchedTaskEnv(B, A, T, device=gpu_device).load_instances(**inst)
    env.reset(seeds, observe=False)
    assert int(env.rollout_random(episodes=1, max_decisions=0).sum()) == 0      # budget 0: nothing happens
    at = np.zeros(B, np.int64)
    launches = 0
    while (at < n).any():
        live = at < n
        budget = live.astype(np.int64)                              # finished envs: budget 0 (they must not start a new episode)
        steps = env.rollout_random(episodes=1, max_decisions=budget).cpu().numpy()
        assert np.array_equal(steps, budget), (launches, steps, budget)
        obs = env.obs()
        ag, tk, mk = (x.cpu().numpy() for x in (obs.agents, obs.tasks, obs.mask))
        for b in np.flatnonzero(live):
            k = int(at[b])
            assert np.array_equal(ag[b], refs[b]["agents_obs"][k]) and np.array_equal(tk[b], refs[b]["tasks_obs"][k]), (b, k)
            assert np.array_equal(mk[b].astype(np.uint8), refs[b]["mask"][k]), (b, k)
        at += budget
        launches += 1
    assert launches == int(n.max())
    fin = H.gpu_final(env)
    for b in range(B):
        H.assert_final_matches(fin[b], refs[b], f"env{b} after {n[b]} single-decision launches")


def test_budget_across_episode_boundaries(gpu_device, golden_dir):
    """Fixed budgets that straddle the auto-reset between episodes reproduce the golden 3-episode run
    (manifest.json multi_episode: reference-generated step counts, rewards and metrics per episode)."""
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.instances import generate_instance
    man = json.load(open(os.path.join(golden_dir, "manifest.json")))["multi_episode"]
    inst = generate_instance(20, 50, man["inst_seed"])
    total = sum(ep["n_steps"] for ep in man["episodes"])
    for chunk in (1000000, 97, 40):
        env = BatchedTaskEnv(1, 20, 50, device=gpu_device)
        env.load_instances(inst["depot"][None], inst["task_xy"][None], inst["req"][None], inst["dur"][None])
        env.reset(np.array([int(man["seed_e"])], np.uint64), observe=False)
        taken, ends, launches = 0, np.cumsum([ep["n_steps"] for ep in man["episodes"]]), 0
        seen = []
        while taken < total:
            b = min(chunk, total - taken)
            got = int(env.rollout_random(episodes=3, max_decisions=b).cpu()[0])
            assert got == b, (chunk, taken, got)
            taken += got
            launches += 1
            if taken in ends:                                       # stopped exactly at an episode end: its results are readable
                seen.append((int(np.flatnonzero(ends == taken)[0]), env.summary().cpu().numpy()[0].copy()))
            assert launches < 1000
        assert int(env.status()["decisions"][0]) == total
        sm = env.summary().cpu().numpy()[0]
        last = man["episodes"][-1]
        assert sm[0] == last["reward"] and int(sm[1]) == last["n_finished"]
        for i in range(6):
            assert sm[2 + i] == last["metrics"][i]
        for idx, s in seen:
            assert s[0] == man["episodes"][idx]["reward"]


def test_rollout_kernel_on_tie_instances(gpu_device, oracle_lib, golden_dir):
    """Symmetric instances (tests/golden/micro_*.npz) produce events with several groups at different locations; the
    persistent kernel must order them like np.unique(axis=0) under the random policy too (checked against the oracle,
    per-decision outputs included at a mid-episode stop)."""
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.choice import env_seeds
    for name in ("micro_ties4", "micro_ties2y", "micro_ties_mixed"):
        tr = H.load_trace(os.path.join(golden_dir, name + ".npz"))
        A, T, B = int(tr["A"]), int(tr["T"]), 16
        env = BatchedTaskEnv(B, A, T, device=gpu_device)
        inst = dict(depot=np.repeat(tr["depot"][None], B, 0), task_xy=np.repeat(tr["task_xy"][None], B, 0),
                    req=np.repeat(tr["req"][None], B, 0), dur=np.repeat(tr["dur"][None], B, 0))
        env.load_instances(**inst)
        seeds = env_seeds(77, 0, B)
        refs = _oracle_traces(oracle_lib, inst, seeds, A, T, B)
        env.reset(seeds, observe=False)
        at = np.array([max(1, r["n_steps"] // 2) for r in refs], np.int64)
        env.rollout_random(1, max_decisions=at)
        _check_last_decision(env, refs, at, name)
        steps = env.rollout_random(1, max_decisions=np.where(at < [r["n_steps"] for r in refs], -1, 0).astype(np.int64)).cpu().numpy()
        fin = H.gpu_final(env)
        for b in range(B):
            assert steps[b] + at[b] == refs[b]["n_steps"], (name, b)
            H.assert_final_matches(fin[b], refs[b], f"{name} env{b}")


def test_ragged_batch_per_decision_outputs(gpu_device, oracle_lib):
    """A ragged batch inside the training range runs the <20,50, runtime sizes> instantiation: per-decision outputs of
    every env against the oracle at that env's own size, padding rows in the policy's convention."""
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.choice import env_seeds
    from dcmrta_amd.instances import generate_batch_ranges
    B = 24
    inst = generate_batch_ranges(range(300, 300 + B), (10, 20), (20, 50))
    A, T = 20, 50
    nA, nT = inst["n_agents"], inst["n_tasks"]
    seeds = env_seeds(8, 0, B)
    env = BatchedTaskEnv(B, A, T, device=gpu_device).load_instances(**inst)
    env.reset(seeds, observe=False)
    refs = []
    for b in range(B):
        a, t = int(nA[b]), int(nT[b])
        o = oracle_lib.OracleEnv(a, t).load(inst["depot"][b], inst["task_xy"][b, :t], inst["req"][b, :t], inst["dur"][b, :t])
        refs.append(o.rollout(int(seeds[b]), 0, oracle_lib.POLICY_RANDOM, record=True))
    rng = np.random.default_rng(3)
    at = rng.integers(1, [r["n_steps"] for r in refs]).astype(np.int64)      # decisions taken: 1 .. n-1
    steps = env.rollout_random(1, max_decisions=at).cpu().numpy()
    assert np.array_equal(steps, at)
    obs = env.obs()
    ag, tk, mk = (x.cpu().numpy() for x in (obs.agents, obs.tasks, obs.mask))
    for b in range(B):
        a, t, k = int(nA[b]), int(nT[b]), int(at[b]) - 1                     # last decision taken
        assert np.array_equal(ag[b, :a], refs[b]["agents_obs"][k]) and np.array_equal(tk[b, :t + 1], refs[b]["tasks_obs"][k]), b
        assert np.array_equal(mk[b, :t + 1].astype(np.uint8), refs[b]["mask"][k]), b
        assert (ag[b, a:] == -1).all() and (tk[b, t + 1:] == -1).all() and mk[b, t + 1:].all(), b
    steps2 = env.rollout_random(1).cpu().numpy()
    sm = env.summary().cpu().numpy()
    for b in range(B):
        assert steps[b] + steps2[b] == refs[b]["n_steps"] and sm[b, 0] == refs[b]["reward"], b


def test_return_log_keeps_every_episode(gpu_device, oracle_lib):
    """dcm_set_return_log: one 3-episode launch leaves all three episode returns in the ring (dcm_summary keeps the last one
    only) -- the per-episode return vector bench.py all-gathers; checked against three one-episode launches of
    a twin handle and, for the first episode, the oracle.  Also through the lockstep API with auto-reset."""
    import torch
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.choice import env_seeds
    from dcmrta_amd.instances import generate_batch
    B, A, T = 24, 20, 50
    inst = generate_batch(B, A, T, base_seed=3)
    seeds = env_seeds(9, 0, B)
    env = BatchedTaskEnv(B, A, T, device=gpu_device).load_instances(**inst)
    ring = env.enable_return_log(3)
    env.reset(seeds, observe=False)
    env.rollout_random(episodes=3)
    twin = BatchedTaskEnv(B, A, T, device=gpu_device).load_instances(**inst)
    twin.reset(seeds, observe=False)
    per_episode = []
    for _ in range(3):
        twin.rollout_random(episodes=1)
        per_episode.append(twin.summary()[:, 0].cpu().numpy().copy())
    got = ring.cpu().numpy()
    for k in range(3):
        assert np.array_equal(got[:, k], per_episode[k]), k
    assert np.array_equal(got[:, 2], env.summary()[:, 0].cpu().numpy())
    for b in range(0, B, 5):
        ref = oracle_lib.OracleEnv(A, T).load(inst["depot"][b], inst["task_xy"][b], inst["req"][b], inst["dur"][b]) \
            .rollout(int(seeds[b]), 0, oracle_lib.POLICY_RANDOM, record=False)
        assert got[b, 0] == ref["reward"]
    # a fourth episode wraps around to column 0
    env.rollout_random(episodes=1)
    assert np.array_equal(ring.cpu().numpy()[:, 0], env.summary()[:, 0].cpu().numpy())
    # lockstep API with auto-reset: the ring fills as episodes end
    ls = BatchedTaskEnv(B, A, T, device=gpu_device, auto_reset=True, auto_reset_episodes=2).load_instances(**inst)
    ring2 = ls.enable_return_log(2)
    obs = ls.reset(seeds)
    for _ in range(600):
        if not bool(obs.active.any()):
            break
        obs = ls.step(torch.argmax((~obs.mask).to(torch.int32), dim=1).int())
    assert not bool(obs.active.any())
    r2 = ring2.cpu().numpy()
    assert not np.isnan(r2).any() and np.array_equal(r2[:, 1], ls.summary()[:, 0].cpu().numpy())


@pytest.mark.parametrize("A,T,B", [(20, 50, 16), (50, 200, 6), (70, 130, 6), (30, 100, 8), (128, 256, 2), (65, 64, 6)])
def test_three_episodes_in_one_launch_against_the_oracle(gpu_device, oracle_lib, A, T, B):
    """One launch plays three consecutive episodes per env (what a bench pass does): the decision counter keeps running across the
    restarts, so episode k of env b is the oracle's rollout from d0 = the decisions of the episodes before it.  Every episode's
    return (dcm_set_return_log), the step total, and the last episode's full terminal state must match -- through each
    register-resident kernel (one chunk, 50A/200T, the mid-size class with one and two agent chunks)."""
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.choice import env_seeds
    from dcmrta_amd.instances import generate_batch
    inst = generate_batch(B, A, T, base_seed=4000 + A + T)
    seeds = env_seeds(17, 0, B)
    env = BatchedTaskEnv(B, A, T, device=gpu_device).load_instances(**inst)
    ring = env.enable_return_log(3)
    env.reset(seeds, observe=False)
    steps = env.rollout_random(episodes=3).cpu().numpy()
    got = ring.cpu().numpy()
    fin = H.gpu_final(env)
    for b in range(B):
        d0, ref = 0, None
        for k in range(3):
            ref = oracle_lib.OracleEnv(A, T).load(inst["depot"][b], inst["task_xy"][b], inst["req"][b], inst["dur"][b]) \
                .rollout(int(seeds[b]), d0, oracle_lib.POLICY_RANDOM, cap_steps=20000, record=False)
            assert got[b, k] == ref["reward"], (b, k, got[b, k], ref["reward"])
            d0 += ref["n_steps"]
        assert steps[b] == d0, (b, steps[b], d0)
        H.assert_final_matches(fin[b], ref, f"{A}A{T}T env{b} third episode")


@pytest.mark.parametrize("mwt", [0.0, 10.0])
def test_zero_max_waiting_time_takes_the_literal_rule(gpu_device, oracle_lib, mwt):
    """max_waiting_time = 0: a member that arrives at the very moment of an event has already "waited" (now - arrival >= 0,
    env/task_env.py:269) and is dropped at once.  The register-resident kernels skip the task_update pass of a quiet join on the
    ground that a fresh member has not waited, which needs max_waiting_time > 0: such a handle runs the general kernels (persistent
    and lockstep), which evaluate the rule literally -- and both match the oracle."""
    import torch
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.choice import env_seeds
    from dcmrta_amd.instances import generate_batch
    B, A, T = 12, 20, 50
    inst = generate_batch(B, A, T, base_seed=321)
    seeds = env_seeds(4, 0, B)
    env = BatchedTaskEnv(B, A, T, device=gpu_device, max_waiting_time=mwt).load_instances(**inst)
    env.reset(seeds, observe=False)
    steps = env.rollout_random(episodes=1).cpu().numpy()
    fin = H.gpu_final(env)
    for b in range(B):
        o = oracle_lib.OracleEnv(A, T, max_waiting_time=mwt).load(inst["depot"][b], inst["task_xy"][b], inst["req"][b], inst["dur"][b])
        ref = o.rollout(int(seeds[b]), 0, oracle_lib.POLICY_RANDOM, cap_steps=50000, record=False)
        assert steps[b] == ref["n_steps"], (b, steps[b], ref["n_steps"])
        H.assert_final_matches(fin[b], ref, f"mwt={mwt} env{b}")
    # the lockstep API under the same parameter: the first 40 decisions of env 0 against the oracle's trace
    obs = env.reset(seeds)
    o = oracle_lib.OracleEnv(A, T, max_waiting_time=mwt).load(inst["depot"][0], inst["task_xy"][0], inst["req"][0], inst["dur"][0])
    ref = o.rollout(int(seeds[0]), 0, oracle_lib.POLICY_FIRST, cap_steps=40, allow_cap=True)
    for i in range(min(40, ref["n_steps"])):
        assert int(obs.leader[0]) == int(ref["leader"][i]) and np.array_equal(obs.mask[0].cpu().numpy().astype(np.uint8), ref["mask"][i]), i
        assert np.array_equal(obs.tasks[0].cpu().numpy().view(np.uint32), ref["tasks_obs"][i].view(np.uint32)), i
        obs = env.step(torch.argmax((~obs.mask).to(torch.int32), dim=1).int())
