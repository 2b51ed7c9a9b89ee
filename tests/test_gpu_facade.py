"""-m gpu: the reference's per-env call pattern (worker.py:45-87) driven against dcmrta_amd.task_env.TaskEnv.

The loop below is the same harness tests/golden/make_golden.py runs against the *reference* TaskEnv (duck-typed:
next_decision / get_unique_group / task_update / agent_update / mask / status / step / check_finished /
get_episode_reward), so equality with the golden traces shows the facade is a drop-in for that call pattern."""
import os

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu
MAX_TIME = 100


def worker_loop(env, seed_e, policy):
    from dcmrta_amd.choice import below, draw
    T = env.tasks_num
    rec = dict(leader=[], action=[], now=[], mask=[], agents_obs=[], tasks_obs=[])
    d = 0
    while not env.finished and env.current_time < MAX_TIME:            # worker.py:45
        decision_agents, current_time = env.next_decision()           # :47
        groups = env.get_unique_group(decision_agents)                # :48
        env.current_time = current_time                               # :49
        env.task_update()                                             # :50
        env.agent_update()                                            # :51
        for group in groups:                                          # :52
            while len(group) > 0:                                     # :53
                leader_id = int(group[below(draw(seed_e, d, 0), len(group))])   # :54 (protocol instead of np.random)
                agent = env.agent_dic[leader_id]
                assert not agent["returned"]                          # :56
                mask = env.get_unfinished_task_mask()                 # :57
                mask = np.insert(mask, 0, False if np.sum(mask) == T else True)  # :58-61
                total_agents = env.get_current_agent_status(agent).astype(np.float32)   # :62
                task_info = env.get_current_task_status(agent).astype(np.float32)       # :64
                action = policy(d, mask)
                rec["leader"].append(leader_id); rec["action"].append(action); rec["now"].append(env.current_time)
                rec["mask"].append(mask.astype(np.uint8)); rec["agents_obs"].append(total_agents); rec["tasks_obs"].append(task_info)
                group, r = env.step(group, leader_id, action, d)      # :73
                env.task_update()                                     # :74
                env.agent_update()                                    # :76
                d += 1
        env.finished = env.check_finished()                           # :85
    reward, finished_tasks = env.get_episode_reward(MAX_TIME)         # :87
    return rec, reward, finished_tasks


@pytest.mark.parametrize("name", ["trace_5A8T_random_s0", "trace_5A8T_nearest_s1", "trace_10A20T_random_s1",
                                  "trace_20A50T_random_s0", "trace_20A50T_nearest_s0", "trace_20A50T_first_s1"])
def test_worker_loop_on_facade(gpu_device, golden_dir, name):
    from dcmrta_amd.task_env import TaskEnv
    tr = H.load_trace(os.path.join(golden_dir, name + ".npz"))
    env = TaskEnv.from_arrays(int(tr["A"]), tr["depot"], tr["task_xy"], tr["req"], tr["dur"], device=gpu_device,
                              choice_seed=int(tr["seed_e"]))
    rec, reward, finished = worker_loop(env, int(tr["seed_e"]), lambda d, mask: int(tr["action"][d]))
    n = int(tr["n_steps"])
    assert len(rec["leader"]) == n
    assert np.array_equal(np.array(rec["leader"]), tr["leader"])
    assert np.array_equal(np.array(rec["now"]), tr["now"])
    assert np.array_equal(np.stack(rec["mask"]), tr["mask"])
    assert np.array_equal(np.stack(rec["agents_obs"]), tr["agents_obs"])
    assert np.array_equal(np.stack(rec["tasks_obs"]), tr["tasks_obs"])
    assert reward == float(tr["reward"])
    assert np.array_equal(np.array(finished, np.uint8), tr["finished"])
    m = env.perf_metrics()
    for i, k in enumerate(("success_rate", "makespan", "time_cost", "waiting_time", "travel_dist", "efficiency")):
        assert m[k] == tr["metrics"][i], k
    # dict views carry the reference's keys
    assert set(("location", "returned", "assigned", "travel_dist")) <= set(env.agent_dic[0])
    assert set(("feasible_assignment", "finished", "time_start", "time_finish", "status", "requirements")) <= set(env.task_dic[0])


@pytest.mark.parametrize("name", ["trace_10A20T_random_s1", "trace_20A50T_random_s0", "trace_20A50T_nearest_s0"])
def test_dict_views_carry_route_and_member_lists(gpu_device, oracle_lib, golden_dir, name):
    """agent_dic[i]['route'] / ['arrival_time'] (env/task_env.py:95-96; read by worker.py:244-251) and
    task_dic[t]['members'] / ['abandoned_agent'] (:78,:89) of the facade against the oracle's own lists after the same
    episode (golden actions; leaders and followers from the shared protocol)."""
    from dcmrta_amd.task_env import TaskEnv
    tr = H.load_trace(os.path.join(golden_dir, name + ".npz"))
    A, T = int(tr["A"]), int(tr["T"])
    env = TaskEnv.from_arrays(A, tr["depot"], tr["task_xy"], tr["req"], tr["dur"], device=gpu_device, choice_seed=int(tr["seed_e"]))
    stop = int(tr["n_steps"]) // 2
    seen_mid = {}

    def act(d, mask):
        if d == stop:      # mid-episode: the views are live, not only terminal
            seen_mid["routes"] = [list(env.agent_dic[a]["route"]) for a in range(A)]
            seen_mid["members"] = [list(env.task_dic[t]["members"]) for t in range(T)]
        return int(tr["action"][d])
    worker_loop(env, int(tr["seed_e"]), act)
    o = oracle_lib.OracleEnv(A, T).load(tr["depot"], tr["task_xy"], tr["req"], tr["dur"])
    ref = o.rollout(int(tr["seed_e"]), 0, oracle_lib.POLICY_INJECTED, inj_action=tr["action"].astype(np.int32))
    assert ref["n_steps"] == int(tr["n_steps"])
    ad, td = env.agent_dic, env.task_dic
    n_ab = 0
    for a in range(A):
        route, arrival = o.route(a)
        assert ad[a]["route"] == route.tolist() and ad[a]["arrival_time"] == arrival.tolist(), a
        assert len(ad[a]["route"]) == ref["route_len"][a]
        assert seen_mid["routes"][a] == route.tolist()[:len(seen_mid["routes"][a])]      # a prefix of the final route
    for t in range(T):
        assert td[t]["members"] == o.members(t).tolist(), t                               # list ORDER matters (quirk Q1)
        assert sorted(td[t]["abandoned_agent"]) == sorted(o.abandoned(t).tolist()), t
        assert len(td[t]["abandoned_agent"]) == ref["n_abandoned"][t]
        n_ab += len(td[t]["abandoned_agent"])
    if "random" in name:
        assert n_ab > 0                                                                   # the random policy does abandon
    assert any(seen_mid["members"])


def test_step_returns_group_and_reward(gpu_device):
    from dcmrta_amd.task_env import TaskEnv
    env = TaskEnv((6, 6), (9, 9), 1, 5, seed=3, device=gpu_device, choice_seed=5)
    ids, t = env.next_decision()
    assert t == 0.0 and list(ids) == list(range(6))
    groups = env.get_unique_group(ids)
    assert groups == [list(range(6))]
    group = groups[0]
    mask = env.get_unfinished_task_mask()
    assert mask.shape == (9,) and not mask.any()
    n0 = len(group)
    group, r = env.step(group, 2, 1, 0)
    assert 2 not in group and len(group) < n0 and r <= 0.0
    with pytest.raises(RuntimeError):
        env.step(group, 2, 1, 0)   # agent 2 is no longer deciding


def test_ctasd_baseline_loop_on_facade(gpu_device, golden_dir):
    """The evaluation loop of baselines/CTAS-D.py:59-94 written against the facade: pre_set_route per agent (CTASD_read_results,
    :36-46), execute_by_route, get_episode_reward, then the script's own metric formulas from current_time / get_matrix(task_dic,
    agent_dic).  Expected values: the reference's per-instance arrays (tests/golden/ctasd_replay.npz); reactive_planning with a
    None route raises TypeError like the reference (env/task_env.py:220)."""
    import copy
    from dcmrta_amd.instances import load_instances_npz, load_routes_json
    from dcmrta_amd.task_env import TaskEnv
    inst, A = load_instances_npz(os.path.join(golden_dir, "instances_20A50T.npz"))
    routes = load_routes_json(os.path.join(golden_dir, "ctasd_routes.json"))
    ref = np.load(os.path.join(golden_dir, "ctasd_replay.npz"))
    idx = ref["idx"].tolist()
    for i in (0, 7, 23):
        env = TaskEnv.from_arrays(A, inst["depot"][i], inst["task_xy"][i], inst["req"][i], inst["dur"][i], device=gpu_device)
        env.reactive_planning = False
        env.clear_decisions()
        for a, r in enumerate(routes[i]):                                    # CTAS-D.py:41-45
            if r == [0]:
                continue
            env.pre_set_route(copy.copy(r)[1:], a)
        env.force_wait = True
        env.execute_by_route("./", "CTAS-D", False)                          # :77
        reward, finished_tasks = env.get_episode_reward(100)                 # :78
        k = idx.index(i)
        assert reward == -ref["makespan"][k] and finished_tasks == [bool(x) for x in ref["finished"][k]]
        assert env.current_time == ref["makespan"][k]                        # :88
        got_tc = np.sum(np.nan_to_num(env.get_matrix(env.task_dic, "time_start"), nan=100))
        assert got_tc == np.sum(np.nan_to_num(ref["time_start"][k], nan=100))                               # :89
        assert np.mean(env.get_matrix(env.agent_dic, "sum_waiting_time")) == np.mean(ref["agent_wait"][k])   # :90
        assert np.sum(env.get_matrix(env.agent_dic, "travel_dist")) == np.sum(ref["travel_dist"][k])         # :91
        assert np.mean(env.get_matrix(env.task_dic, "sum_waiting_time")) == np.mean(ref["task_wait"][k])    # :92
    rr = np.load(os.path.join(golden_dir, "reactive_replay.npz"))
    raised = rr["raised"].tolist()
    assert raised, "fixture should contain instances where the reference raises"
    i = raised[0]
    env = TaskEnv.from_arrays(A, inst["depot"][i], inst["task_xy"][i], inst["req"][i], inst["dur"][i], device=gpu_device)
    env.reactive_planning = True
    for a, r in enumerate(routes[i]):
        if r != [0]:
            env.pre_set_route(copy.copy(r)[1:], a)
    with pytest.raises(TypeError):
        env.execute_by_route("./", "CTAS-D", False)


def test_reset_with_reference_dicts(gpu_device, golden_dir):
    """env.reset((task_dic, agent_dic, depot)) with dicts laid out like the reference's (env/task_env.py:76-113), as
    RL_test.py:36-42 does after unpickling: the instance is swapped in and the episode equals the one of from_arrays."""
    from dcmrta_amd.instances import load_instances_npz
    from dcmrta_amd.task_env import TaskEnv
    inst, A = load_instances_npz(os.path.join(golden_dir, "instances_20A50T.npz"))
    i = 3
    task_dic = {t: {"ID": t, "requirements": np.array([inst["req"][i][t]]), "members": [], "location": inst["task_xy"][i][t].copy(),
                    "time": np.array([inst["dur"][i][t]]), "status": np.array([inst["req"][i][t]])} for t in range(50)}
    agent_dic = {a: {"ID": a, "location": inst["depot"][i].copy(), "route": []} for a in range(A)}
    depot = {"location": inst["depot"][i].copy(), "members": [], "ID": -1}

    def episode(env):
        n = 0
        while not env.finished and env.current_time < 100:
            ids, t = env.next_decision()
            groups = env.get_unique_group(ids)
            env.current_time = t
            for group in groups:
                while group:
                    leader = group[0]
                    m = env.get_unfinished_task_mask()
                    action = int(np.flatnonzero(~m)[0]) + 1 if not m.all() else 0
                    group, _ = env.step(group, leader, action, n)
                    n += 1
            env.finished = env.check_finished()
        return n, env.get_episode_reward(100)[0]

    a = TaskEnv((5, 5), (7, 7), 1, 5, seed=1, device=gpu_device)           # some other instance first
    a.reset((task_dic, agent_dic, depot))
    assert (a.agents_num, a.tasks_num) == (A, 50) and np.array_equal(a.depot["location"], inst["depot"][i])
    b = TaskEnv.from_arrays(A, inst["depot"][i], inst["task_xy"][i], inst["req"][i], inst["dur"][i], device=gpu_device)
    assert episode(a) == episode(b)


def test_committing_a_foreign_time_is_refused(gpu_device):
    """task_update / agent_update are no-ops of the facade (the device applied them at its own event time): a caller that sets
    another current_time -- after which the reference would recompute from that time, env/task_env.py:245-281 -- gets an
    exception, not stale answers."""
    from dcmrta_amd.task_env import TaskEnv
    env = TaskEnv((5, 5), (8, 8), seed=3, device=gpu_device)
    ids, t = env.next_decision()
    env.current_time = t                      # the reference's call order (worker.py:49): fine
    env.task_update(); env.agent_update()
    with pytest.raises(ValueError, match="only the event time"):
        env.current_time = t + 1.25
    assert env.current_time == t
