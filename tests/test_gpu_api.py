"""-m gpu: behaviour of the C-ABI surface itself -- purity of observe, snapshot/restore (copy.deepcopy of worker.py:33),
per-env error flags, inactive envs, maximum sizes, size-independent properties at BASELINE's full sizes."""
import numpy as np
import pytest
import torch

import helpers as H

pytestmark = pytest.mark.gpu


def _mk(B, A, T, dev, seed=0, **kw):
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.instances import generate_batch
    inst = generate_batch(B, A, T, base_seed=seed)
    return BatchedTaskEnv(B, A, T, device=dev, **kw).load_instances(**inst), inst


def _oracle_random(oracle_lib, inst, seeds, A, T, b, record):
    o = oracle_lib.OracleEnv(A, T).load(inst["depot"][b], inst["task_xy"][b], inst["req"][b], inst["dur"][b])
    return o.rollout(int(seeds[b]), 0, oracle_lib.POLICY_RANDOM, record=record)


def test_observe_is_pure_and_matches_fused_observe(gpu_device, oracle_lib):
    """dcm_observe is a pure function of the state: it returns what the fused observe of dcm_step wrote, twice -- and both
    are the oracle's record of that decision (protocol-random actions chosen on the host)."""
    from dcmrta_amd.choice import env_seeds
    B, A, T = 32, 20, 50
    env, inst = _mk(B, A, T, gpu_device)
    seeds = env_seeds(1, 0, B)
    refs = [_oracle_random(oracle_lib, inst, seeds, A, T, b, True) for b in range(B)]
    obs = env.reset(seeds)
    for d in range(15):
        a = [t.clone() for t in (obs.agents, obs.tasks, obs.mask, obs.leader, obs.active)]
        for again in (env.observe(), env.observe()):
            for x, y in zip(a, (again.agents, again.tasks, again.mask, again.leader, again.active)):
                assert torch.equal(x, y)
        ag, tk, mk, ld = (t.cpu().numpy() for t in a[:4])
        for b in range(B):
            r = refs[b]
            assert ld[b] == r["leader"][d] and np.array_equal(ag[b], r["agents_obs"][d]) and np.array_equal(tk[b], r["tasks_obs"][d])
            assert np.array_equal(mk[b].astype(np.uint8), r["mask"][d])
        act = np.array([H.host_random_action(mk[b].astype(np.uint8), int(seeds[b]), d) for b in range(B)], np.int32)
        assert all(act[b] == refs[b]["action"][d] for b in range(B))
        obs = env.step(act)


def test_clone_restore_replays_identically(gpu_device, oracle_lib):
    """Greedy-twin use of worker.py:33,89: snapshot mid-episode, play the episode out, restore, play it out again -> the same
    episode both times, and it is the oracle's episode (the first 7 decisions through the lockstep API, the rest in the
    persistent kernel, all from the shared choice protocol)."""
    from dcmrta_amd.choice import env_seeds
    B, A, T = 64, 20, 50
    env, inst = _mk(B, A, T, gpu_device, seed=5)
    seeds = env_seeds(2, 0, B)
    refs = [_oracle_random(oracle_lib, inst, seeds, A, T, b, False) for b in range(B)]
    obs = env.reset(seeds)
    for d in range(7):
        mk = obs.mask.cpu().numpy().astype(np.uint8)
        obs = env.step(np.array([H.host_random_action(mk[b], int(seeds[b]), d) for b in range(B)], np.int32))
    snap = env.clone_state()
    s1 = env.rollout_random(1).clone()
    r1 = env.summary().clone()
    f1 = H.gpu_final(env)
    env.restore_state(snap)
    assert (env.status()["decisions"] == 7).all()
    s2 = env.rollout_random(1)
    assert torch.equal(s1, s2) and torch.equal(r1, env.summary())
    f2 = H.gpu_final(env)
    for b in range(B):
        assert int(s1[b]) + 7 == refs[b]["n_steps"], b
        H.assert_final_matches(f1[b], refs[b], f"env{b} first run")
        H.assert_final_matches(f2[b], refs[b], f"env{b} after restore")


def test_error_flags_freeze_only_the_offending_env(gpu_device):
    from dcmrta_amd import _lib
    from dcmrta_amd.choice import env_seeds
    env, _ = _mk(4, 6, 9, gpu_device)
    obs = env.reset(env_seeds(3, 0, 4))
    act = torch.ones(4, dtype=torch.int32, device=gpu_device)
    act[1] = 99       # out of range -> BAD_ACTION
    act[2] = -3
    obs = env.step(act)
    flags = env.status()["flags"].cpu().numpy()
    assert flags[1] & _lib.FLAG_BAD_ACTION and flags[2] & _lib.FLAG_BAD_ACTION and flags[1] & _lib.FLAG_DONE
    assert flags[0] == 0 and flags[3] == 0
    assert obs.active.cpu().tolist() == [True, False, False, True]
    assert obs.leader[1] == -1 and obs.mask[1].cpu().tolist() == [False] + [True] * 9   # inactive rows: only the depot unmasked
    # injected leader that is not deciding
    obs2 = env.observe(leader=np.array([0, -1, -1, 0], np.int32))
    valid = torch.argmax((~obs2.mask).to(torch.int32), dim=1).int()                  # first unmasked action per env
    env.step(valid, leader=np.array([-1, -1, -1, 5], np.int32), observe=False)
    # agent 5 decided at t=0 unless it already left with a previous leader; either it moved or BAD_LEADER is raised
    st = env.status()["flags"].cpu().numpy()
    assert st[3] in (0, _lib.FLAG_BAD_LEADER | _lib.FLAG_DONE)
    nan_rows = torch.isnan(env.summary()[:, 0]).cpu().numpy()
    assert nan_rows.all()   # nobody finished an episode yet -> summary rows are NaN


def test_masked_task_action_is_refused_in_strict_mode(gpu_device):
    """DCM_PARAM_STRICT_MASK: a host-supplied action on a task the mask forbids (worker.py:57-61) sets BAD_ACTION and freezes
    that env only.  (Without the flag the action is simulated like TaskEnv.step does: test_masked_actions_are_simulated.)"""
    from dcmrta_amd import _lib
    from dcmrta_amd.choice import env_seeds
    env, _ = _mk(3, 6, 9, gpu_device, strict_mask=True)
    obs = env.reset(env_seeds(5, 0, 3))
    one = torch.ones(3, dtype=torch.int32, device=gpu_device)
    obs = env.step(one)                                  # task 0 gets exactly its requirement -> status 0 -> masked
    assert obs.mask[:, 1].all() and not env.status()["flags"].cpu().numpy().any()
    act = one.clone()
    unmasked = [int(np.flatnonzero(~obs.mask[b].cpu().numpy())[0]) for b in range(3)]
    act[0], act[2] = unmasked[0], unmasked[2]            # env 1 repeats the now-masked action
    obs = env.step(act)
    flags = env.status()["flags"].cpu().numpy()
    assert flags[1] == (_lib.FLAG_BAD_ACTION | _lib.FLAG_DONE) and flags[0] == 0 and flags[2] == 0
    assert obs.active.cpu().tolist() == [True, False, True]


def test_handles_of_different_shapes_coexist(gpu_device, oracle_lib):
    """The dynamic-LDS limit is a per-kernel attribute shared by all handles: creating a small env after a large one
    must not break the large one's launches (runtime-shape kernels, 100A/300T needs > 64 KiB of LDS)."""
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.choice import env_seeds
    from dcmrta_amd.instances import generate_batch
    big_inst, small_inst = generate_batch(2, 100, 300, base_seed=5), generate_batch(2, 5, 8, base_seed=5)
    big = BatchedTaskEnv(2, 100, 300, device=gpu_device).load_instances(**big_inst)
    small = BatchedTaskEnv(2, 5, 8, device=gpu_device).load_instances(**small_inst)
    seeds = env_seeds(1, 0, 2)
    for env, inst, (A, T) in ((big, big_inst, (100, 300)), (small, small_inst, (5, 8)), (big, big_inst, (100, 300))):
        env.reset(seeds, observe=False)
        steps = env.rollout_random(1).cpu().numpy()
        fin = H.gpu_final(env)
        for b in range(2):
            ref = oracle_lib.OracleEnv(A, T).load(inst["depot"][b], inst["task_xy"][b], inst["req"][b], inst["dur"][b]).rollout(
                int(seeds[b]), 0, oracle_lib.POLICY_RANDOM, cap_steps=100000, record=False)
            assert steps[b] == ref["n_steps"]
            H.assert_final_matches(fin[b], ref, f"{A}A{T}T env{b}")


def test_many_abandonments_stay_exact(gpu_device, oracle_lib):
    """More than 16 abandonments of one agent in an episode (needs max_waiting_time << MAX_TIME, here 3 vs 250) overflow
    the per-agent abandonment log; such agents are then summed from the dense per-(agent, task) count table, in task
    order like env/task_env.py:358-364 -- every per-agent waiting sum stays bit-exact and DCM_FLAG_WAIT_ORDER (a counter
    saturated at 255) does not appear."""
    from dcmrta_amd import _lib
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.choice import env_seeds
    from dcmrta_amd.instances import generate_batch
    B, A, T = 32, 20, 63
    inst = generate_batch(B, A, T, base_seed=77)
    seeds = env_seeds(8, 0, B)
    most = 0
    for mwt, mt in ((3.0, 250.0), (10.0, 100.0)):
        env = BatchedTaskEnv(B, A, T, device=gpu_device, max_waiting_time=mwt, max_time=mt).load_instances(**inst)
        env.reset(seeds, observe=False)
        steps = env.rollout_random(2).cpu().numpy()                      # two episodes: the tables are cleared at the restart
        fin = H.gpu_final(env)
        for b in range(B):
            o = oracle_lib.OracleEnv(A, T, max_waiting_time=mwt, max_time=mt).load(
                inst["depot"][b], inst["task_xy"][b], inst["req"][b], inst["dur"][b])
            r1 = o.rollout(int(seeds[b]), 0, oracle_lib.POLICY_RANDOM, cap_steps=100000, record=False)
            o.clear_decisions()
            ref = o.rollout(int(seeds[b]), r1["n_steps"], oracle_lib.POLICY_RANDOM, cap_steps=100000, record=False)
            assert steps[b] == r1["n_steps"] + ref["n_steps"]
            assert not fin[b]["flags"] & _lib.FLAG_WAIT_ORDER
            H.assert_final_matches(fin[b], ref, f"mwt={mwt} env{b}")
            if mwt == 3.0:
                most = max(most, int(ref["n_abandoned"].sum()))
    assert most > 16 * 3          # the scenario really exercises the overflow path


def test_api_state_errors(gpu_device):
    from dcmrta_amd import _lib
    from dcmrta_amd.batched_env import BatchedTaskEnv
    env = BatchedTaskEnv(2, 3, 4, device=gpu_device)
    with pytest.raises(_lib.DcmError):
        env.reset(np.zeros(2, np.uint64))            # no instances loaded
    with pytest.raises(_lib.DcmError):
        env.load_instances(np.zeros((2, 2)), np.zeros((2, 4, 2)), np.full((2, 4), 6), np.ones((2, 4)))  # requirement > 5
    with pytest.raises(_lib.DcmError):
        BatchedTaskEnv(2, 300, 4, device=gpu_device)  # A > DCM_MAX_AGENTS


@pytest.mark.parametrize("A,T", [(128, 1023), (128, 64), (1, 1023), (64, 1)])
def test_limit_sizes_match_oracle(gpu_device, oracle_lib, A, T):
    from dcmrta_amd.choice import env_seeds
    env, inst = _mk(2, A, T, gpu_device, seed=70)
    seeds = env_seeds(5, 0, 2)
    env.reset(seeds, observe=False)
    steps = env.rollout_random(1).cpu().numpy()
    fin = H.gpu_final(env)
    for b in range(2):
        o = oracle_lib.OracleEnv(A, T).load(inst["depot"][b], inst["task_xy"][b], inst["req"][b], inst["dur"][b])
        ref = o.rollout(int(seeds[b]), 0, oracle_lib.POLICY_RANDOM, cap_steps=50000, record=False)
        assert steps[b] == ref["n_steps"]
        H.assert_final_matches(fin[b], ref, f"{A}A{T}T env{b}")


def test_full_size_properties(gpu_device):
    """BASELINE full sizes (4096 x 20A/50T, 8192 x 50A/200T): properties that do not need the oracle."""
    from dcmrta_amd.choice import env_seeds
    for B, A, T in ((4096, 20, 50), (8192, 50, 200)):
        env, inst = _mk(B, A, T, gpu_device)
        seeds = env_seeds(0, 0, B)
        env.reset(seeds, observe=False)
        s1 = env.rollout_random(1).clone()
        sm1 = env.summary().clone()
        ts = {k: v.clone() for k, v in env.tasks_state().items()}
        ag = {k: v.clone() for k, v in env.agents_state().items()}
        # determinism: same seeds -> identical episode
        env.reset(seeds, observe=False)
        assert torch.equal(env.rollout_random(1), s1) and torch.equal(env.summary(), sm1)
        sm = sm1.cpu().numpy()
        assert np.isfinite(sm).all() and (env.status()["flags"].cpu().numpy() & 0x7C).sum() == 0
        fin, feas = ts["finished"].cpu().numpy().astype(bool), ts["feasible"].cpu().numpy().astype(bool)
        assert not (fin & ~feas).any()                                   # finished => feasible
        assert np.array_equal(fin.sum(1), sm[:, 1].astype(int))          # n_finished is the checksum of the flags
        assert np.allclose(sm[:, 2], fin.mean(1)) and np.array_equal(sm[:, 0], -sm[:, 3])
        nm, req = ts["n_members"].cpu().numpy(), inst["req"]
        assert (nm <= req).all() and (nm[feas] == req[feas]).all()       # coalition size: <= requirement, == when formed
        tstart, tfin, dur = ts["time_start"].cpu().numpy(), ts["time_finish"].cpu().numpy(), inst["dur"]
        assert np.array_equal(tfin[feas], (tstart + dur)[feas])
        assert np.array_equal(ag["travel_dist"].sum(1).cpu().numpy() > 0, np.ones(B, bool))
        # a makespan beyond MAX_TIME only through quirk Q7 (the event that crosses 100 is processed completely)
        assert (sm[:, 3] < 100 + 10 + 5 * 2 ** 0.5 + 5).all()
        assert int(s1.min()) > 0
        # the same episodes cut into launches of 37 decisions (every launch reloads the record and writes it back; at 50A/200T the
        # persistent kernel keeps the member arrival times in the HBM record itself) and, in between, one lockstep-API observe
        env.reset(seeds, observe=False)
        total = torch.zeros_like(s1)
        for it in range(64):
            done = (env.status()["flags"] & 1).bool() if it else torch.zeros_like(s1, dtype=torch.bool)
            if bool(done.all()):
                break
            # (an env whose episode is over would start the next one: budget 0 leaves it where it is)
            total += env.rollout_random(1, max_decisions=torch.where(done, 0, 37).to(torch.int64))
            if it == 3:
                env.observe()
        assert torch.equal(total, s1) and torch.equal(env.summary(), sm1)
        ts2 = env.tasks_state()
        assert all(torch.equal(ts2[k], v) for k, v in ts.items())


@pytest.mark.parametrize("A,T,policy", [(8, 14, "first"), (20, 50, "random"), (3, 40, "random"), (33, 9, "random"), (70, 70, "last")])
def test_individual_selection_mode(gpu_device, oracle_lib, A, T, policy):
    """Worker.run_test_IS (worker.py:159-198): every deciding agent acts alone, in ascending id order, without
    get_unique_group.  Reference loop restated on the oracle's step-wise surface; the action is a deterministic function
    of the mask (first / last / keyed-random valid action) so both sides take the same decisions."""
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.instances import generate_batch
    B = 4
    inst = generate_batch(B, A, T, base_seed=61)

    def choose(mask_row, b, i):
        valid = np.flatnonzero(np.asarray(mask_row) == 0)
        if policy == "first":
            return int(valid[0])
        if policy == "last":
            return int(valid[-1])
        return int(valid[(1103515245 * (i + 7 * b + 1) + 12345) % len(valid)])

    env = BatchedTaskEnv(B, A, T, device=gpu_device, individual_selection=True).load_instances(**inst)
    env.reset(np.arange(B, dtype=np.uint64), observe=False)
    # device: always inject the lowest pending agent as leader, no followers
    steps = np.zeros(B, int)
    for _ in range(5000):
        pg = env.agents_state()["pending_group"].cpu().numpy()
        flags = env.status()["flags"].cpu().numpy()
        if (flags & 1).all():
            break
        lead = np.array([int(np.flatnonzero(pg[b] > 0)[0]) if not (flags[b] & 1) else -1 for b in range(B)], np.int32)
        obs = env.observe(leader=lead)
        mk = obs.mask.cpu().numpy()
        act = torch.tensor([choose(mk[b], b, int(steps[b])) if not (flags[b] & 1) else 0 for b in range(B)], dtype=torch.int32)
        env.step(act, leader=lead, n_followers=np.zeros(B, np.int32), followers=np.full((B, 4), -1, np.int16), observe=False)
        steps += (~(flags & 1).astype(bool)).astype(int)
    fin = H.gpu_final(env)
    for b in range(B):
        o = oracle_lib.OracleEnv(A, T).load(inst["depot"][b], inst["task_xy"][b], inst["req"][b], inst["dur"][b])
        n, finished, guard = 0, False, 0
        while not finished and o.now < 100:                                  # worker.py:163
            ids, t = o.next_decision()                                       # :165
            o.now = t                                                        # :167
            o.task_update(); o.agent_update()                                # :168-169
            for a in ids:                                                    # :170
                m = o.mask()                                                 # :175-179
                o.agent_step(int(a), choose(m, b, n))                        # :185-186
                o.task_update(); o.agent_update()                            # :187-188
                n += 1
            finished = o.check_finished()                                    # :189
            guard += 1
            assert guard < 5000
        ref = o.final()
        from oracle import lib as _ol
        _ol().orc_finish_episode(o._h)
        ref = o.final()
        assert steps[b] == n, (b, steps[b], n)
        H.assert_final_matches(fin[b], ref, f"IS env{b}")


def test_auto_reset_lockstep_matches_multi_episode_golden(gpu_device, golden_dir, oracle_lib):
    """DCM_PARAM_AUTO_RESET: dcm_step restarts an env in the call that ends its episode (consecutive episodes,
    auto-reset to the same instance, decision counter keeps running).  B = 1 against the reference-generated 3-episode golden
    (manifest.json multi_episode), then a batch against the oracle episode by episode (d0 = decisions so far)."""
    import json
    import os
    import helpers as H
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.choice import env_seeds
    from dcmrta_amd.instances import generate_batch, generate_instance
    man = json.load(open(os.path.join(golden_dir, "manifest.json")))["multi_episode"]
    inst = generate_instance(20, 50, man["inst_seed"])
    seed = int(man["seed_e"])
    env = BatchedTaskEnv(1, 20, 50, device=gpu_device, auto_reset=True)
    env.load_instances(inst["depot"][None], inst["task_xy"][None], inst["req"][None], inst["dur"][None])
    obs = env.reset(np.array([seed], np.uint64))
    d, done = 0, 0
    bounds = np.cumsum([ep["n_steps"] for ep in man["episodes"]])
    while done < 3:
        assert bool(obs.active[0])                                   # the env never goes idle
        a = H.host_random_action(obs.mask[0].cpu().numpy().astype(np.uint8), seed, d)
        obs = env.step(np.array([a], np.int32))
        d += 1
        n_ep = int(env.episodes()[0])
        if n_ep > done:
            ep = man["episodes"][done]
            assert d == bounds[done], (done, d, bounds[done])
            sm = env.summary().cpu().numpy()[0]
            assert sm[0] == ep["reward"] and int(sm[1]) == ep["n_finished"]
            for i in range(6):
                assert sm[2 + i] == ep["metrics"][i]
            done = n_ep
            assert float(env.status()["now"][0]) == 0.0              # the new episode sits at its first event
    # batch: every env against the oracle, episode after episode
    B, A, T = 24, 12, 23
    insts = generate_batch(B, A, T, base_seed=40)
    seeds = env_seeds(6, 0, B)
    env = BatchedTaskEnv(B, A, T, device=gpu_device, auto_reset=True).load_instances(**insts)
    obs = env.reset(seeds)
    refs = []
    for b in range(B):
        o = oracle_lib.OracleEnv(A, T).load(insts["depot"][b], insts["task_xy"][b], insts["req"][b], insts["dur"][b])
        r1 = o.rollout(int(seeds[b]), 0, oracle_lib.POLICY_RANDOM, record=False)
        o.clear_decisions()
        r2 = o.rollout(int(seeds[b]), r1["n_steps"], oracle_lib.POLICY_RANDOM, record=False)
        refs.append((r1, r2))
    dcount = np.zeros(B, np.int64)
    seen = np.zeros(B, np.int64)
    for _ in range(2000):
        mk = obs.mask.cpu().numpy().astype(np.uint8)
        act = np.array([H.host_random_action(mk[b], int(seeds[b]), int(dcount[b])) for b in range(B)], np.int32)
        obs = env.step(act)
        dcount += 1
        eps = env.episodes().cpu().numpy()
        sm = env.summary().cpu().numpy()
        for b in np.flatnonzero(eps > seen):
            k = int(seen[b])
            if k < 2:
                ref = refs[b][k]
                total = refs[b][0]["n_steps"] + (refs[b][1]["n_steps"] if k == 1 else 0)
                assert dcount[b] == total and sm[b, 0] == ref["reward"], (b, k)
                for i in range(6):
                    assert sm[b, 2 + i] == ref["metrics"][i], (b, k, i)
        seen = np.maximum(seen, eps)
        if (seen >= 2).all():
            break
    assert (seen >= 2).all() and bool(obs.active.all())


def test_auto_reset_with_individual_selection(gpu_device):
    """DCM_PARAM_AUTO_RESET | DCM_PARAM_NO_GROUPING: individual selection draws nothing from the choice protocol (lowest pending
    id decides, alone), so under a deterministic policy every episode of an env repeats the first one exactly -- which is the
    episode the same env plays without auto-reset (itself checked against the reference loop in
    test_individual_selection_mode)."""
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.choice import env_seeds
    from dcmrta_amd.instances import generate_batch
    B, A, T = 12, 9, 17
    inst = generate_batch(B, A, T, base_seed=3)
    seeds = env_seeds(4, 0, B)
    first_valid = lambda obs: torch.argmax((~obs.mask).to(torch.int32), dim=1).int()
    ref = BatchedTaskEnv(B, A, T, device=gpu_device, individual_selection=True).load_instances(**inst)
    obs = ref.reset(seeds)
    while bool(obs.active.any()):
        obs = ref.step(first_valid(obs))
    want, n1 = ref.summary().clone(), ref.status()["decisions"].clone()
    env = BatchedTaskEnv(B, A, T, device=gpu_device, individual_selection=True, auto_reset=True,
                         auto_reset_episodes=3).load_instances(**inst)
    obs = env.reset(seeds)
    seen = torch.zeros(B, dtype=torch.int32, device=gpu_device)
    for _ in range(5000):
        if not bool(obs.active.any()):
            break
        obs = env.step(first_valid(obs))
        eps = env.episodes()
        newly = eps > seen
        if bool(newly.any()):
            assert torch.equal(env.summary()[newly], want[newly])          # every finished episode equals the first one
            seen = eps.clone()
    assert (env.episodes() == 3).all() and not bool(obs.active.any())
    assert torch.equal(env.status()["decisions"], 3 * n1) and torch.equal(env.summary(), want)


def test_requirements_are_validated_on_the_device(gpu_device):
    """The C ABI itself refuses a requirement outside 1..DCM_MAX_MEMBERS (the Python wrapper checks too, but a maintainer binds
    the ABI): the env is flagged by dcm_load_instances and never starts; its neighbours are untouched."""
    import ctypes as C
    import torch
    from dcmrta_amd import _lib
    from dcmrta_amd.batched_env import BatchedTaskEnv, _ptr
    from dcmrta_amd.choice import env_seeds
    from dcmrta_amd.instances import generate_batch
    B, A, T = 6, 5, 8
    inst = generate_batch(B, A, T, base_seed=5)
    req = inst["req"].copy()
    req[2, 3] = 7
    req[4, 0] = 0
    env = BatchedTaskEnv(B, A, T, device=gpu_device)
    dev = env.device
    d = torch.as_tensor(inst["depot"]).to(dev); xy = torch.as_tensor(inst["task_xy"]).to(dev)
    rq = torch.as_tensor(req, dtype=torch.int32).to(dev); du = torch.as_tensor(inst["dur"]).to(dev)
    _lib.check(env._lib.dcm_load_instances(env._h, _ptr(d), _ptr(xy), _ptr(rq), _ptr(du), env._stream()))
    obs = env.reset(env_seeds(1, 0, B))
    flags = env.status()["flags"].cpu().numpy()
    bad = np.array([False, False, True, False, True, False])
    assert ((flags & _lib.FLAG_BAD_INSTANCE) != 0).tolist() == bad.tolist()
    assert (obs.active.cpu().numpy() == ~bad).all()
    steps = env.rollout_random(episodes=1).cpu().numpy()
    assert (steps[bad] == 0).all() and (steps[~bad] > 0).all()
    flags = env.status()["flags"].cpu().numpy()
    assert ((flags & _lib.FLAG_BAD_INSTANCE) != 0).tolist() == bad.tolist()
    # the valid envs played exactly what they play in a clean batch
    clean = BatchedTaskEnv(B, A, T, device=gpu_device).load_instances(**inst)
    clean.reset(env_seeds(1, 0, B), observe=False)
    ref = clean.rollout_random(episodes=1).cpu().numpy()
    assert (steps[~bad] == ref[~bad]).all()
    assert np.array_equal(env.summary().cpu().numpy()[~bad], clean.summary().cpu().numpy()[~bad])
    # a valid instance clears the mark
    env.load_instances(**inst)
    env.reset(env_seeds(1, 0, B), observe=False)
    assert not (env.status()["flags"].cpu().numpy() & _lib.FLAG_BAD_INSTANCE).any()


def _anymask_action(mask_row, seed_e, d, T):
    """Host mirror of ORC_POLICY_ANY / tests/golden/make_golden_masked.py: a policy that does not respect the mask."""
    from dcmrta_amd.choice import below, draw
    r = draw(seed_e, d, 1)
    if r % 16 == 1:
        return 0
    if r % 4 == 0:
        return 1 + (r >> 4) % T
    valid = np.flatnonzero(mask_row == 0)
    return int(valid[below(r, len(valid))])


def test_masked_actions_are_simulated(gpu_device, oracle_lib, golden_dir):
    """TaskEnv.step has no mask check (env/task_env.py:326-342): an action on a feasible / full / stale-status task sends the
    leader alone, the task lists a surplus member, an agent that joins a task that is already over is released in the past
    and the event time steps backwards.  dcm_step restates all of that; the one limit is the 5 member slots per task: the
    episode freezes with DCM_FLAG_OVERFLOW at exactly the decision whose agent_step would make a sixth member.
    (The reference-generated traces trace_*_anymask_* run through test_gpu_parity's golden tests like every other trace.)"""
    import glob
    import os
    from dcmrta_amd import _lib
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.choice import env_seeds
    from dcmrta_amd.instances import generate_batch
    # (1) a batch of random instances, mask-ignoring policy on the host, every env against the oracle
    B, A, T = 64, 12, 25
    inst = generate_batch(B, A, T, base_seed=400)
    seeds = env_seeds(13, 0, B)
    env = BatchedTaskEnv(B, A, T, device=gpu_device).load_instances(**inst)
    got = H.run_lockstep(env, seeds, lambda b, i, m, l: _anymask_action(m, int(seeds[b]), i, T))
    fin = H.gpu_final(env)
    n_exact = n_over = masked = backwards = 0
    for b in range(B):
        o = oracle_lib.OracleEnv(A, T).load(inst["depot"][b], inst["task_xy"][b], inst["req"][b], inst["dur"][b])
        ref = o.rollout(int(seeds[b]), 0, oracle_lib.POLICY_ANY, cap_steps=20000)
        overflowed = bool(fin[b]["flags"] & _lib.FLAG_OVERFLOW)
        assert overflowed == (ref["max_members_seen"] > 5), b            # the freeze is the slot limit, nothing else
        n = got[b]["n_steps"]
        for k in ("leader", "action", "now", "mask", "agents_obs", "tasks_obs"):   # identical up to the freeze / the end
            assert np.array_equal(got[b][k], ref[k][:n]), (b, k)
        if overflowed:
            n_over += 1
            continue
        assert n == ref["n_steps"] and not fin[b]["flags"] & _lib.FLAG_BAD_ACTION
        H.assert_final_matches(fin[b], ref, f"env{b}")
        n_exact += 1
        masked += int(sum(ref["mask"][i][a] for i, a in enumerate(ref["action"])))
        backwards += int((np.diff(ref["now"]) < 0).sum())
    assert n_exact >= 20 and masked > 100 and backwards > 5, (n_exact, n_over, masked, backwards)
    # (2) the reference's own episodes that run into the slot limit: equal up to that decision, frozen there
    for p in sorted(glob.glob(os.path.join(golden_dir, "overflow_*.npz"))):
        tr = H.load_trace(p)
        A2, T2 = int(tr["A"]), int(tr["T"])
        e2 = BatchedTaskEnv(1, A2, T2, device=gpu_device).load_instances(tr["depot"][None], tr["task_xy"][None], tr["req"][None],
                                                                          tr["dur"][None])
        g = H.run_lockstep(e2, np.array([int(tr["seed_e"])], np.uint64), lambda b, i, m, l: int(tr["action"][i]))[0]
        stop = int(tr["overflow_step"])
        assert g["n_steps"] == stop + 1                                     # the overflowing decision is the last one taken
        for k in ("leader", "now", "mask", "agents_obs", "tasks_obs"):
            assert np.array_equal(g[k], tr[k][:stop + 1]), (p, k)
        assert int(e2.status()["flags"].cpu().numpy()[0]) & _lib.FLAG_OVERFLOW
