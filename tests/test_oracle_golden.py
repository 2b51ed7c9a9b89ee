"""CPU: pins the oracle (oracle/dcmrta_oracle.c) against the reference.

 (i)  reference-published known answer: the CTAS-D routes replayed through execute_by_route reproduce
      testSet_20A_50T_CONDET/metrics/metrics.csv:2 (the only numbers the reference ships for this path);
 (ii) golden vectors produced by importing the reference (tests/golden/make_golden.py): RL-mode step traces,
      route-replay outputs with and without dynamic task visibility, distance known answers.
"""
import json
import os

import numpy as np
import pytest

import helpers as H

POLICY = {"random": 0, "first": 2, "nearest": 3, "anymask": 4}   # anymask: ignores the mask (make_golden_masked.py)
ALL_KEYS = ("leader", "action", "nfol", "followers", "now", "mask", "agents_obs", "tasks_obs", "finished", "feasible",
            "time_start", "time_finish", "task_wait", "n_members", "n_abandoned", "agent_wait", "travel_dist",
            "returned", "route_len", "metrics")


@pytest.mark.parametrize("path", H.full_traces(), ids=os.path.basename)
def test_full_trace_bit_exact(oracle_lib, path):
    tr = H.load_trace(path)
    A, T = int(tr["A"]), int(tr["T"])
    pol = H.trace_policy(path)
    e = oracle_lib.OracleEnv(A, T).load(tr["depot"], tr["task_xy"], tr["req"], tr["dur"])
    out = e.rollout(int(tr["seed_e"]), 0, POLICY[pol], cap_steps=4096)
    assert out["n_steps"] == int(tr["n_steps"])
    assert out["reward"] == float(tr["reward"])
    assert out["truncated"] == int(tr["truncated"])
    for k in ALL_KEYS:
        assert np.array_equal(np.asarray(out[k]), tr[k]), k


@pytest.mark.parametrize("path", [p for p in H.full_traces() if "20A50T" in p or "5A8T" in p or "micro_" in p], ids=os.path.basename)
def test_injected_replay_equals_protocol(oracle_lib, path):
    """Replaying the recorded (leader, followers, action) by injection gives the same episode as drawing them."""
    tr = H.load_trace(path)
    A, T = int(tr["A"]), int(tr["T"])
    e = oracle_lib.OracleEnv(A, T).load(tr["depot"], tr["task_xy"], tr["req"], tr["dur"])
    out = e.rollout(0xDEAD, 0, oracle_lib.POLICY_INJECTED, cap_steps=4096, inj_leader=tr["leader"],
                    inj_action=tr["action"], inj_nfol=tr["nfol"], inj_followers=tr["followers"])
    for k in ALL_KEYS:
        assert np.array_equal(np.asarray(out[k]), tr[k]), k


def test_hashed_traces(oracle_lib):
    from dcmrta_amd.instances import generate_instance
    for name, meta in H.trace_hashes().items():
        A, T = meta["A"], meta["T"]
        inst = generate_instance(A, T, meta["inst_seed"])
        e = oracle_lib.OracleEnv(A, T).load(inst["depot"], inst["task_xy"], inst["req"], inst["dur"])
        out = e.rollout(int(meta["seed_e"]), 0, POLICY[meta["policy"]], cap_steps=8192)
        assert out["n_steps"] == meta["n_steps"], name
        assert H.digest(out) == meta["sha256"], name


def extra_instance(meta):
    """Instance of an entry of trace_hashes_extra.json (tests/golden/make_golden_extra.py)."""
    from dcmrta_amd.instances import generate_instance, generate_instance_ranges
    if meta["kind"] == "ranges":
        A, inst = generate_instance_ranges((10, 20), (20, 50), meta["inst_seed"])
    elif meta["kind"] == "arrays":           # hand-modified instance (coincident locations): arrays stored with the digest
        A = meta["A"]
        inst = dict(depot=np.array(meta["depot"]), task_xy=np.array(meta["task_xy"]), req=np.array(meta["req"], np.int32),
                    dur=np.array(meta["dur"]))
    else:
        A = meta["A"]
        inst = generate_instance(A, meta["T"], meta["inst_seed"], meta.get("max_coalition_size", 5), meta.get("max_duration", 5.0))
    assert (A, len(inst["req"])) == (meta["A"], meta["T"])
    assert int(inst["req"].sum()) == meta["req_sum"] and float(inst["dur"][0]) == meta["dur0"]
    return A, inst


def test_extra_hashed_traces(oracle_lib, golden_dir):
    """Reference digests on the axes the first golden set keeps fixed: sizes drawn from (10,20)x(20,50) ranges,
    max_waiting_time 3 / 25, loop bound 30 / 250, coalition size 3, durations 2 / 0, more agents than tasks, tasks with
    coincident coordinates / on the depot (one group per LOCATION, env/task_env.py:291-298)."""
    hashes = json.load(open(os.path.join(golden_dir, "trace_hashes_extra.json")))
    assert len(hashes) >= 42
    for name, meta in hashes.items():
        A, inst = extra_instance(meta)
        e = oracle_lib.OracleEnv(A, meta["T"], max_waiting_time=meta.get("max_waiting_time", 10.0),
                                 max_time=meta.get("max_time", 100.0)).load(inst["depot"], inst["task_xy"], inst["req"], inst["dur"])
        out = e.rollout(int(meta["seed_e"]), 0, POLICY[meta["policy"]], cap_steps=8192)
        assert out["n_steps"] == meta["n_steps"] and out["reward"] == meta["reward"], name
        assert int(out["finished"].sum()) == meta["n_finished"], name
        assert H.digest(out) == meta["sha256"], name


def test_testset_instances_rl_mode(oracle_lib, golden_dir):
    """The 50 shipped instances have per-task durations drawn U(0,5) (older generator): RL-mode digests."""
    from dcmrta_amd.instances import load_instances_npz
    inst, A = load_instances_npz(os.path.join(golden_dir, "instances_20A50T.npz"))
    hashes = json.load(open(os.path.join(golden_dir, "testset_rl_hashes.json")))
    for i, meta in hashes.items():
        i = int(i)
        e = oracle_lib.OracleEnv(A, 50).load(inst["depot"][i], inst["task_xy"][i], inst["req"][i], inst["dur"][i])
        out = e.rollout(int(meta["seed_e"]), 0, POLICY[meta["policy"]], cap_steps=4096)
        assert out["n_steps"] == meta["n_steps"] and out["reward"] == meta["reward"]
        assert H.digest(out) == meta["sha256"]


REPLAY_KEYS = ("makespan", "metrics", "finished", "time_start", "time_finish", "task_wait", "agent_wait", "travel_dist",
               "returned", "n_members", "route_len")


def _replay(oracle_lib, golden_dir, i, reactive):
    from dcmrta_amd.instances import load_instances_npz, load_routes_json
    inst, A = load_instances_npz(os.path.join(golden_dir, "instances_20A50T.npz"))
    routes = load_routes_json(os.path.join(golden_dir, "ctasd_routes.json"))
    e = oracle_lib.OracleEnv(A, 50).load(inst["depot"][i], inst["task_xy"][i], inst["req"][i], inst["dur"][i])
    for a, r in enumerate(routes[i]):  # baselines/CTAS-D.py:41-45
        if r == [0]:
            continue
        e.pre_set_route(r[1:], a)
    return e.execute_by_route(reactive)


@pytest.mark.parametrize("reactive,name", [(False, "ctasd_replay"), (True, "reactive_replay")])
def test_route_replay_bit_exact(oracle_lib, golden_dir, reactive, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    idx, raised = list(g["idx"]), list(g["raised"])
    for i in range(50):
        if i in raised:  # env/task_env.py:220 raises TypeError in the reference (pre_set_route is None)
            with pytest.raises(TypeError):
                _replay(oracle_lib, golden_dir, i, reactive)
            continue
        out = _replay(oracle_lib, golden_dir, i, reactive)
        k = idx.index(i)
        for key in REPLAY_KEYS:
            assert np.array_equal(np.asarray(out[key]), g[key][k]), (i, key)


def test_published_known_answer(oracle_lib, golden_dir):
    """metrics/metrics.csv:2 -- 'CTAS-D_300s,1.0 (+- 0.0),36.908 (+- 3.754),0.0,5.619 (+- 1.767),42.027 (+- 4.492),
    2.248 (+- 0.707)' (mean, population std over the 50 instances, rounded to 3 decimals by results_plotting.py)."""
    m = np.stack([_replay(oracle_lib, golden_dir, i, False)["metrics"] for i in range(50)])
    mean, std = m.mean(0), m.std(0)
    assert np.all(m[:, 0] == 1.0)
    assert round(mean[1], 3) == 36.908 and round(std[1], 3) == 3.754
    assert round(mean[3], 3) == 5.619 and round(std[3], 3) == 1.767
    assert round(mean[4], 3) == 42.027 and round(std[4], 3) == 4.492
    assert round(mean[5], 3) == 2.248 and round(std[5], 3) == 0.707


def test_distance_kat(oracle_lib, golden_dir):
    """G5 through the oracle's agent_step: one agent, one task -> travel_dist == np.linalg.norm of the reference."""
    z = np.load(os.path.join(golden_dir, "distance_kat.npz"))
    for i in range(0, 4096, 7):
        e = oracle_lib.OracleEnv(1, 1).load(z["a"][i], z["b"][i][None, :], np.array([1]), np.array([5.0]))
        e.agent_step(0, 1)
        assert e.final()["travel_dist"][0] == z["dist"][i]


def test_pairwise_sum_equals_numpy(oracle_lib):
    rng = np.random.default_rng(3)
    for n in list(range(0, 40)) + [50, 100, 127, 128, 129, 200, 255, 256, 257, 500, 1000, 1023]:
        a = rng.random(n) * 100
        assert oracle_lib.pairwise_sum(a) == float(np.sum(a)), n


def test_choice_protocol_mirrors(oracle_lib):
    from dcmrta_amd import choice
    for z in (0, 1, 12345, 2**63 + 7, 2**64 - 1):
        assert oracle_lib.mix64(z) == choice.mix64(z)
    for b, e in ((0, 0), (1000, 3), (2**40, 65535)):
        assert oracle_lib.env_seed(b, e) == choice.env_seed(b, e)
        assert choice.env_seeds(b, e, 2)[0] == choice.env_seed(b, e)
        s = choice.env_seed(b, e)
        for d in (0, 1, 119, 10**6):
            for slot in (0, 1, 2, 3, 5, 9):
                assert oracle_lib.draw(s, d, slot) == choice.draw(s, d, slot)


def test_quirks_are_exercised(golden_dir):
    q = json.load(open(os.path.join(golden_dir, "manifest.json")))["quirks_in_traces"]
    assert q["Q1_skip_after_removal"] > 0 and q["Q2_stale_member_decides"] > 0 and q["Q3_stale_status_masked"] > 0
    assert q["Q4_rejoin"] > 0 and q["Q7_over_max_time"] > 0 and q["multi_group_events"] > 0


def test_micro_ties_have_several_groups(oracle_lib, golden_dir):
    """micro_ties4: four agents finish four symmetric tasks at the same instant -> one event, four groups in
    ascending (x, y) order (np.unique(axis=0), env/task_env.py:293)."""
    tr = H.load_trace(os.path.join(golden_dir, "micro_ties4.npz"))
    e = oracle_lib.OracleEnv(4, 8).load(tr["depot"], tr["task_xy"], tr["req"], tr["dur"])
    for a, act in enumerate((1, 2, 3, 4)):          # tasks at (0.75,0.5), (0.5,0.75), (0.25,0.5), (0.5,0.25)
        e.agent_step(a, act)
    e.task_update(); e.agent_update()
    ids, t = e.next_decision()
    assert list(ids) == [0, 1, 2, 3] and t == 0.25 / 0.2 + 5.0
    assert e.get_unique_group(ids) == [[2], [3], [1], [0]]


def random_replay_cases(golden_dir):
    """tests/golden/replay_random.json: (case, instance dict) pairs; instances from the test set or the seeded generator."""
    from dcmrta_amd.instances import generate_instance, load_instances_npz
    cases = json.load(open(os.path.join(golden_dir, "replay_random.json")))
    ts, _ = load_instances_npz(os.path.join(golden_dir, "instances_20A50T.npz"))
    out = []
    for c in cases:
        if c["kind"] == "testset":
            inst = {k: ts[k][c["index"]] for k in ("depot", "task_xy", "req", "dur")}
        else:
            inst = generate_instance(c["A"], c["T"], c["inst_seed"])
        assert int(inst["req"].sum()) == c["req_sum"]
        out.append((c, inst))
    return out


def test_random_route_replays(oracle_lib, golden_dir):
    """execute_by_route of the REFERENCE on random routes (surplus visitors that are released before they arrive, too few
    visitors, pre_set_route None, shuffled visiting order; tests/golden/make_golden_extra.py): every terminal array bit
    for bit, TypeError where the reference raises it, and the zero-decider guard where the reference never terminates."""
    cases = random_replay_cases(golden_dir)
    seen = {"ok": 0, "type_error": 0, "no_termination": 0}
    for c, inst in cases:
        o = oracle_lib.OracleEnv(c["A"], c["T"]).load(inst["depot"], inst["task_xy"], inst["req"], inst["dur"])
        for a, r in enumerate(c["routes"]):
            if r is not None:
                o.pre_set_route(r, a)
        try:
            ref = o.execute_by_route(c["reactive"])
            status = "no_termination" if ref["truncated"] else "ok"
        except TypeError:
            status = "type_error"
        assert status == c["status"], (c["A"], c["T"], c["reactive"])
        seen[status] += 1
        if status == "ok":
            for k, v in c["result"].items():
                exp = np.asarray(v)
                assert np.array_equal(np.asarray(ref[k]).astype(exp.dtype), exp, equal_nan=True), (c["A"], c["T"], c["reactive"], k)
    assert seen["ok"] >= 15 and seen["type_error"] >= 10 and seen["no_termination"] >= 4


def schedule_cases(golden_dir):
    """tests/golden/replay_schedule.json (make_golden_schedule.py): replays of the reference with its four dynamic-arrival
    constants substituted (env/task_env.py:567, :221) -> (case, instance dict) pairs."""
    from dcmrta_amd.instances import generate_instance
    out = []
    for c in json.load(open(os.path.join(golden_dir, "replay_schedule.json"))):
        inst = generate_instance(c["A"], c["T"], c["inst_seed"])
        assert int(inst["req"].sum()) == c["req_sum"]
        out.append((c, inst))
    return out


def test_generalised_visibility_schedule(oracle_lib, golden_dir):
    """The parametrised schedule (orc_set_visibility) against the reference run with the same four constants: every terminal
    array bit for bit, incl. a schedule under which all 500 tasks of a 100A/500T instance become visible, the reference's own
    constants (an identity check of the harness), TypeError cases and one the reference never finishes."""
    seen = {"ok": 0, "type_error": 0, "no_termination": 0}
    for c, inst in schedule_cases(golden_dir):
        o = oracle_lib.OracleEnv(c["A"], c["T"]).load(inst["depot"], inst["task_xy"], inst["req"], inst["dur"])
        o.set_visibility(*c["schedule"])
        for a, r in enumerate(c["routes"]):
            if r is not None:
                o.pre_set_route(r, a)
        try:
            ref = o.execute_by_route(True)
            status = "no_termination" if ref["truncated"] else "ok"
        except TypeError:
            status = "type_error"
        assert status == c["status"], (c["A"], c["T"], c["schedule"])
        seen[status] += 1
        if status == "ok":
            for k, v in c["result"].items():
                exp = np.asarray(v)
                assert np.array_equal(np.asarray(ref[k]).astype(exp.dtype), exp, equal_nan=True), (c["A"], c["T"], c["schedule"], k)
    assert seen["ok"] >= 6


def test_visibility_schedule_validation(oracle_lib):
    o = oracle_lib.OracleEnv(3, 4)
    for bad in ((-1, 20, 10, 100), (20, 0, 10, 100), (20, 20, 0, 100), (20, 20, 10, 19)):
        with pytest.raises(ValueError):
            o.set_visibility(*bad)


def test_mask_ignoring_policy_traces(oracle_lib, golden_dir):
    """tests/golden/make_golden_masked.py: the reference driven by a policy that picks masked tasks (and the depot while tasks
    remain) -- TaskEnv.step simulates those (env/task_env.py:326-342).  The full traces trace_*_anymask_* are checked by
    test_full_trace_bit_exact like every other trace; here: they really contain masked picks and backward time steps, and the
    episodes in which a task lists more than 5 members (overflow_*: the HIP env stops there) match to the very end as well."""
    import glob
    traces = sorted(glob.glob(os.path.join(golden_dir, "trace_*_anymask_*.npz")))
    assert len(traces) >= 10
    assert sum(int(np.load(p)["masked_picks"]) for p in traces) > 50
    assert sum(int(np.load(p)["time_steps_backwards"]) for p in traces) > 5
    over = sorted(glob.glob(os.path.join(golden_dir, "overflow_*.npz")))
    assert len(over) >= 3
    for p in over:
        tr = H.load_trace(p)
        A, T = int(tr["A"]), int(tr["T"])
        e = oracle_lib.OracleEnv(A, T).load(tr["depot"], tr["task_xy"], tr["req"], tr["dur"])
        out = e.rollout(int(tr["seed_e"]), 0, oracle_lib.POLICY_ANY, cap_steps=4096)
        assert out["n_steps"] == int(tr["n_steps"]) and out["reward"] == float(tr["reward"])
        for k in ALL_KEYS:
            assert np.array_equal(np.asarray(out[k]), tr[k]), (p, k)
        assert int(tr["n_members"].max()) > 5 or int(tr["overflow_step"]) >= 0
