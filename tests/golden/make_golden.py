#!/usr/bin/env python3
"""Golden-vector generator (runs ONLY in the build container, never on the GPU box).

Imports the reference simulator read-only from /root/reference and drives it with
*injected* choices (the keyed splitmix64 protocol of DESIGN.md §"Choice protocol"),
then writes numbers-only fixtures under tests/golden/:

  instances_20A50T.npz   G1  the 50 test-set instances (depot, task xy, requirement, duration)
  ctasd_routes.json      G2  CTAS-D routes parsed like baselines/CTAS-D.py:10-33
  ctasd_replay.npz       G3  execute_by_route outputs per instance (known answer:
                             aggregates == testSet_20A_50T_CONDET/metrics/metrics.csv:2)
  reactive_replay.npz    G6  same with reactive_planning=True (dynamic task visibility)
  distance_kat.npz       G5  np.linalg.norm known answers for 2-vectors
  trace_*.npz            G4  RL-mode step traces (full tensors)
  trace_hashes.json      G4  sha256 digests of further traces
  manifest.json          counts of quirk conditions Q1-Q4,Q7 hit by the committed traces

Usage (from any cwd that is not /root/reference):
  PYTHONPATH=/root/reference MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 \
      python3 tests/golden/make_golden.py

Nothing from the reference (source, bytecode, pickles) is copied: fixtures hold inputs
and expected outputs only.  Reference lines restated by the harness loop are cited inline.
"""
import copy
import hashlib
import json
import os
import pickle
import sys
import warnings

import numpy as np

warnings.filterwarnings("ignore")
REF = os.environ.get("DCMRTA_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True

from env.task_env import TaskEnv  # noqa: E402  (reference, read-only)
import __main__  # noqa: E402

__main__.TaskEnv = TaskEnv  # the test-set pickles reference __main__.TaskEnv (RL_test.py:6)

OUT = os.path.dirname(os.path.abspath(__file__))
MAX_TIME = 100  # parameters.py:18
M64 = (1 << 64) - 1
GAMMA = 0x9E3779B97F4A7C15


# ----------------------------------------------------------------------------- choice protocol
def mix64(z):
    z &= M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def env_seed(base, e):
    return mix64(base + GAMMA * (e + 1))


def draw(seed_e, d, s):
    """32-bit word s of the decision's stream hi(key_1), lo(key_1), hi(key_2), ... (dcmrta_amd/choice.py)."""
    key = mix64(seed_e + GAMMA * (d + 1))
    for _ in range(s // 2):
        key = mix64(key + GAMMA)
    return (key >> 32) if s % 2 == 0 else (key & 0xFFFFFFFF)


def below(r, n):
    return (r * n) >> 32


# ----------------------------------------------------------------------------- instance helpers
def instance_arrays(env):
    T, A = env.tasks_num, env.agents_num
    depot = np.asarray(env.depot["location"], dtype=np.float64).copy()
    xy = np.stack([np.asarray(env.task_dic[i]["location"], dtype=np.float64) for i in range(T)])
    req = np.array([int(np.asarray(env.task_dic[i]["requirements"]).reshape(-1)[0]) for i in range(T)], dtype=np.int32)
    dur = np.array([float(np.asarray(env.task_dic[i]["time"]).reshape(-1)[0]) for i in range(T)], dtype=np.float64)
    return dict(depot=depot, task_xy=xy, req=req, dur=dur, A=np.int32(A), T=np.int32(T))


def load_testset_env(i, max_waiting_time=10):
    """RL_test.py:34-44 / baselines/CTAS-D.py:59-67."""
    env = pickle.load(open(f"{REF}/testSet_20A_50T_CONDET/env_{i}.pkl", "rb"))
    env.max_waiting_time = max_waiting_time
    env.reactive_planning = False
    env.visible_length = 0
    env.reset((env.task_dic, env.agent_dic, env.depot))
    env.clear_decisions()
    return env


# ----------------------------------------------------------------------------- policies
def policy_random(env, mask, leader, seed_e, d):
    valid = np.flatnonzero(~mask)
    return int(valid[below(draw(seed_e, d, 1), len(valid))])


def policy_first(env, mask, leader, seed_e, d):
    return int(np.flatnonzero(~mask)[0])


def policy_nearest(env, mask, leader, seed_e, d):
    valid = np.flatnonzero(~mask)
    if valid[0] == 0:
        return 0
    loc = env.agent_dic[leader]["location"]
    best, bd = None, None
    for a in valid:
        t = env.task_dic[a - 1]["location"]
        dd = float(np.linalg.norm(loc - t))
        if bd is None or dd < bd:
            best, bd = int(a), dd
    return best


POLICIES = {"random": policy_random, "first": policy_first, "nearest": policy_nearest}


# ----------------------------------------------------------------------------- RL-mode harness
def rollout(env, seed_e, policy, d0=0, record=True, quirks=None):
    """Restates the loop of worker.py:45-87 with injected leader/follower/action choices."""
    A, T = env.agents_num, env.tasks_num
    state = {"d": d0}

    def injected_choice(a, size=None, replace=True):  # shadows env/task_env.py:50-55
        rest = list(a)
        k = int(np.asarray(size).reshape(-1)[0])
        out = []
        for j in range(k):
            out.append(rest.pop(below(draw(seed_e, state["d"], 2 + j), len(rest))))
        return np.array(out, dtype=np.int64)

    env.random_choice = injected_choice
    rec = {k: [] for k in ("leader", "action", "nfol", "followers", "now", "mask", "agents", "tasks")}
    empty_passes = 0
    truncated = False
    while not env.finished and env.current_time < MAX_TIME:  # worker.py:45
        ids, t = env.next_decision()  # :47
        groups = env.get_unique_group(ids) if len(ids) else []  # :48
        env.current_time = t  # :49
        if quirks is not None:
            _count_q1(env, quirks)
            if len(groups) > 1:
                quirks["multi_group_events"] = quirks.get("multi_group_events", 0) + 1
        env.task_update()  # :50
        env.agent_update()  # :51
        if len(groups) == 0:
            empty_passes += 1
            if empty_passes > 4:  # guard: the reference would spin forever here
                truncated = True
                break
        else:
            empty_passes = 0
        for group in groups:  # :52
            while len(group) > 0:  # :53
                d = state["d"]
                leader = int(group[below(draw(seed_e, d, 0), len(group))])  # :54, injected
                agent = env.agent_dic[leader]
                assert not agent["returned"]  # :56
                m = env.get_unfinished_task_mask()  # :57
                m = np.insert(m, 0, False if np.sum(m) == T else True)  # :58-61
                ag = np.asarray(env.get_current_agent_status(agent), dtype=np.float64)  # :62
                tk = np.asarray(env.get_current_task_status(agent), dtype=np.float64)  # :64
                action = policy(env, m, leader, seed_e, d)
                before = list(group)
                if quirks is not None:
                    _count_quirks_pre(env, action, leader, quirks)
                group, r = env.step(group, leader, action, d)  # :73
                moved = [x for x in before if x not in group]
                followers = [x for x in moved if x != leader]
                # recover follower ORDER as drawn (members = [leader] + followers)
                followers = _follower_order(seed_e, d, before, leader, len(followers))
                env.task_update()  # :74
                env.agent_update()  # :76
                if record:
                    rec["leader"].append(leader)
                    rec["action"].append(action)
                    rec["nfol"].append(len(followers))
                    f = np.full(A, -1, dtype=np.int16)
                    f[: len(followers)] = followers
                    rec["followers"].append(f)
                    rec["now"].append(float(env.current_time))
                    rec["mask"].append(m.astype(np.uint8))
                    rec["agents"].append(ag.astype(np.float32))  # FloatTensor cast, worker.py:62
                    rec["tasks"].append(tk.astype(np.float32))  # worker.py:64
                state["d"] += 1
        env.finished = env.check_finished()  # :85
    reward, finished_tasks = env.get_episode_reward(MAX_TIME)  # :87
    out = final_arrays(env)
    out["reward"] = np.float64(reward)
    out["truncated"] = np.uint8(truncated)
    out["n_steps"] = np.int64(state["d"] - d0)
    if record:
        n = len(rec["leader"])
        out["leader"] = np.array(rec["leader"], dtype=np.int32)
        out["action"] = np.array(rec["action"], dtype=np.int32)
        out["nfol"] = np.array(rec["nfol"], dtype=np.int32)
        out["followers"] = np.stack(rec["followers"]) if n else np.zeros((0, A), np.int16)
        out["now"] = np.array(rec["now"], dtype=np.float64)
        out["mask"] = np.stack(rec["mask"]) if n else np.zeros((0, T + 1), np.uint8)
        out["agents_obs"] = np.stack(rec["agents"]) if n else np.zeros((0, A, 6), np.float32)
        out["tasks_obs"] = np.stack(rec["tasks"]) if n else np.zeros((0, T + 1, 5), np.float32)
    return out


def _follower_order(seed_e, d, before, leader, k):
    rest = [x for x in before if x != leader]
    out = []
    for j in range(k):
        out.append(rest.pop(below(draw(seed_e, d, 2 + j), len(rest))))
    return out


def _count_q1(env, q):
    """Q1 (env/task_env.py:268-271): two consecutive members expire in the same update -> the second one is skipped."""
    for t in env.task_dic.values():
        if t["feasible_assignment"] or len(t["members"]) < 2:
            continue
        if int(np.asarray(t["requirements"]).reshape(-1)[0]) - len(t["members"]) <= 0:
            continue
        ex = [env.current_time - env.get_arrival_time(m, t["ID"]) >= env.max_waiting_time for m in t["members"]]
        if any(a and b for a, b in zip(ex, ex[1:])):
            q["Q1_skip_after_removal"] = q.get("Q1_skip_after_removal", 0) + 1


def _count_quirks_pre(env, action, leader, q):
    # Q4 rejoin: the chosen task already lists the leader as a member (env/task_env.py:321)
    if action > 0 and leader in env.task_dic[action - 1]["members"]:
        q["Q4_rejoin"] += 1
    # Q2 stale member: leader decides while still counted in some task's members (:269)
    for t in env.task_dic.values():
        if (not t["feasible_assignment"]) and leader in t["members"]:
            q["Q2_stale_member_decides"] += 1
            break
    # Q3 stale status: infeasible task with status<=0 (masked for this decision) (:260-265)
    for t in env.task_dic.values():
        if (not t["feasible_assignment"]) and int(np.asarray(t["status"]).reshape(-1)[0]) <= 0:
            q["Q3_stale_status_masked"] += 1
            break


def final_arrays(env):
    """Terminal outputs: worker.py:87,103-108 after get_episode_reward()."""
    A, T = env.agents_num, env.tasks_num
    td, ad = env.task_dic, env.agent_dic
    fin = np.array([bool(td[i]["finished"]) for i in range(T)], dtype=np.uint8)
    out = dict(
        makespan=np.float64(env.current_time),
        finished=fin,
        feasible=np.array([bool(td[i]["feasible_assignment"]) for i in range(T)], dtype=np.uint8),
        time_start=np.array([float(td[i]["time_start"]) for i in range(T)], dtype=np.float64),
        time_finish=np.array([float(td[i]["time_finish"]) for i in range(T)], dtype=np.float64),
        task_wait=np.array([float(td[i]["sum_waiting_time"]) for i in range(T)], dtype=np.float64),
        n_members=np.array([len(td[i]["members"]) for i in range(T)], dtype=np.int32),
        n_abandoned=np.array([len(td[i]["abandoned_agent"]) for i in range(T)], dtype=np.int32),
        agent_wait=np.array([float(ad[i]["sum_waiting_time"]) for i in range(A)], dtype=np.float64),
        travel_dist=np.array([float(ad[i]["travel_dist"]) for i in range(A)], dtype=np.float64),
        returned=np.array([bool(ad[i]["returned"]) for i in range(A)], dtype=np.uint8),
        route_len=np.array([len(ad[i]["route"]) for i in range(A)], dtype=np.int32),
    )
    # the six perf metrics of worker.py:103-108
    out["metrics"] = np.array(
        [
            np.sum(fin) / len(fin),
            env.current_time,
            np.nanmean(env.get_matrix(td, "time_start")),
            np.mean(env.get_matrix(ad, "sum_waiting_time")),
            np.sum(env.get_matrix(ad, "travel_dist")),
            np.mean(env.get_matrix(td, "sum_waiting_time")),
        ],
        dtype=np.float64,
    )
    return out


def digest(tr):
    h = hashlib.sha256()
    for k in ("leader", "action", "nfol", "followers", "now", "mask", "agents_obs", "tasks_obs",
              "metrics", "finished", "time_start", "travel_dist", "agent_wait", "task_wait"):
        h.update(np.ascontiguousarray(tr[k]).tobytes())
    return h.hexdigest()


# ----------------------------------------------------------------------------- route replay
def ctasd_routes(i):
    """baselines/CTAS-D.py:10-33 without yaml dependency on the param file (vehNum == 20)."""
    import yaml

    p = f"{REF}/testSet_20A_50T_CONDET/env_{i}/"
    with open(p + "planner_param.yaml") as f:
        pd = yaml.safe_load(f)
    num_veh = pd["vehNum"] if pd["flagSolver"] == "TEAMPLANNER_DET" else pd["vehNumPerType"][0]
    with open(p + "results.yaml") as f:
        data = yaml.safe_load(f)
    if "vehicle" not in data:
        return None
    nodes = []
    for v in range(num_veh):
        if "vv" + str(v + 1) not in data["vehicle"]:
            continue
        nodes.append(list(data["vehicle"]["vv" + str(v + 1)]["node"]))
    return nodes


def replay(i, routes, reactive):
    env = load_testset_env(i)
    env.reactive_planning = reactive
    for a, r in enumerate(routes):  # baselines/CTAS-D.py:41-45
        if r == [0]:
            continue
        env.pre_set_route(copy.copy(r)[1:], a)
    env.force_wait = True
    import contextlib
    import io

    with contextlib.redirect_stdout(io.StringIO()):
        env.execute_by_route("./", "CTAS-D", False)  # env/task_env.py:562-593
    env.get_episode_reward(100)
    return env


def main():
    os.makedirs(OUT, exist_ok=True)
    manifest = {}

    # ---------------------------------------------------------------- G1 instances
    inst = [instance_arrays(load_testset_env(i)) for i in range(50)]
    np.savez_compressed(
        f"{OUT}/instances_20A50T.npz",
        depot=np.stack([x["depot"] for x in inst]),
        task_xy=np.stack([x["task_xy"] for x in inst]),
        req=np.stack([x["req"] for x in inst]),
        dur=np.stack([x["dur"] for x in inst]),
        A=np.int32(20),
    )

    # ---------------------------------------------------------------- G2 + G3 CTAS-D replay
    routes_all = {}
    keys = ("makespan", "metrics", "finished", "time_start", "time_finish", "task_wait", "agent_wait",
            "travel_dist", "returned", "n_members", "route_len")
    for reactive, name in ((False, "ctasd_replay"), (True, "reactive_replay")):
        rows = {k: [] for k in keys}
        ok, raised = [], []
        for i in range(50):
            routes = ctasd_routes(i)
            routes_all[str(i)] = routes
            try:
                env = replay(i, routes, reactive)
            except TypeError as ex:  # env/task_env.py:220 with pre_set_route None (SURVEY a-16)
                raised.append(i)
                continue
            fa = final_arrays(env)
            for k in keys:
                rows[k].append(fa[k])
            ok.append(i)
        np.savez_compressed(f"{OUT}/{name}.npz", idx=np.array(ok, np.int32), raised=np.array(raised, np.int32),
                            **{k: np.stack(v) for k, v in rows.items()})
        m = np.stack(rows["metrics"])
        manifest[name] = dict(n_ok=len(ok), raised=raised,
                              mean=dict(zip(("success", "makespan", "time_cost", "waiting", "travel", "efficiency"),
                                            [float(x) for x in m.mean(0)])),
                              std=dict(zip(("success", "makespan", "time_cost", "waiting", "travel", "efficiency"),
                                           [float(x) for x in m.std(0, ddof=1)])))
    with open(f"{OUT}/ctasd_routes.json", "w") as f:
        json.dump(routes_all, f, separators=(",", ":"))

    # ---------------------------------------------------------------- G5 distance KAT
    rng = np.random.default_rng(12345)
    a = rng.random((4096, 2))
    b = rng.random((4096, 2))
    dist = np.array([np.linalg.norm(a[i] - b[i]) for i in range(len(a))], dtype=np.float64)
    np.savez_compressed(f"{OUT}/distance_kat.npz", a=a, b=b, dist=dist)
    dx, dy = a[:, 0] - b[:, 0], a[:, 1] - b[:, 1]
    import math
    fma_form = np.array([math.sqrt(_fma(dy[i], dy[i], dx[i] * dx[i])) for i in range(len(a))])
    plain = np.sqrt(dx * dx + dy * dy)
    manifest["distance_kat"] = dict(n=len(a), equals_fma_form=int(np.sum(fma_form == dist)),
                                    equals_plain_form=int(np.sum(plain == dist)))

    # ---------------------------------------------------------------- G4 RL traces
    quirks = dict(Q4_rejoin=0, Q2_stale_member_decides=0, Q3_stale_status_masked=0, Q7_over_max_time=0)
    full = []  # (name, A, T, policy, base_seed)
    for (A, T) in ((5, 8), (10, 20), (20, 50)):
        for pol in ("random", "nearest", "first"):
            for s in (0, 1):
                full.append((A, T, pol, s))
    full += [(50, 200, "random", 0), (50, 200, "nearest", 0)]
    hashes = {}
    hashed = [(20, 50, "random", s) for s in range(2, 22)] + [(20, 50, "nearest", s) for s in range(2, 12)] \
        + [(50, 200, "random", s) for s in range(1, 4)] + [(100, 500, "random", 0), (100, 500, "nearest", 0)] \
        + [(13, 37, "random", s) for s in range(4)] + [(64, 64, "random", 0), (70, 130, "random", 0)]
    for (A, T, pol, s) in full + hashed:
        env = TaskEnv((A, A), (T, T), 1, 5, seed=s)  # worker.py:32 with fixed sizes; generate_env :57-114
        ia = instance_arrays(env)
        se = env_seed(1000 + s, 0)
        tr = rollout(env, se, POLICIES[pol], quirks=quirks)
        if tr["makespan"] >= MAX_TIME:
            quirks["Q7_over_max_time"] += 1
        name = f"trace_{A}A{T}T_{pol}_s{s}"
        if (A, T, pol, s) in full:
            np.savez_compressed(f"{OUT}/{name}.npz", seed_e=np.uint64(se), inst_seed=np.int64(s), **ia, **tr)
        else:
            hashes[name] = dict(A=A, T=T, policy=pol, inst_seed=s, seed_e=str(se), sha256=digest(tr),
                                n_steps=int(tr["n_steps"]), reward=float(tr["reward"]),
                                n_finished=int(tr["finished"].sum()))
        print(name, int(tr["n_steps"]), float(tr["reward"]), int(tr["finished"].sum()), flush=True)
    with open(f"{OUT}/trace_hashes.json", "w") as f:
        json.dump(hashes, f, indent=1)

    # G6 micro-scenario: exact time ties at DIFFERENT locations -> several groups per event, ordered like the rows of
    # np.unique(axis=0) (env/task_env.py:291-298).  Symmetric instance: four requirement-1 tasks at distance 0.25 from
    # the depot, so the four agents sent there alone arrive and finish at identical times.
    for name, A, xy, req in (
            ("ties4", 4, [(0.75, 0.5), (0.5, 0.75), (0.25, 0.5), (0.5, 0.25), (0.1, 0.9), (0.9, 0.1), (0.3, 0.3), (0.7, 0.7)], [1] * 8),
            ("ties2y", 2, [(0.5, 0.75), (0.5, 0.25), (0.2, 0.2), (0.8, 0.8), (0.8, 0.2)], [1, 1, 2, 1, 1]),
            ("ties_mixed", 6, [(0.25, 0.5), (0.75, 0.5), (0.5, 0.25), (0.5, 0.75), (0.25, 0.25), (0.75, 0.75), (0.1, 0.5)],
             [2, 2, 1, 1, 3, 1, 2])):
        T = len(req)
        env = TaskEnv((A, A), (T, T), 1, 5, seed=0)
        dep = np.array([0.5, 0.5])
        for i in range(T):
            env.task_dic[i]["location"] = np.array(xy[i], dtype=np.float64)
            env.task_dic[i]["requirements"] = np.array([req[i]])
            env.task_dic[i]["status"] = np.array([req[i]])
        env.depot["location"] = dep
        for a in env.agent_dic.values():
            a["location"] = dep
            a["depot"] = dep
        env.clear_decisions()
        ia = instance_arrays(env)
        se = env_seed(4000, T)
        tr = rollout(env, se, POLICIES["first"], quirks=quirks)
        np.savez_compressed(f"{OUT}/micro_{name}.npz", seed_e=np.uint64(se), inst_seed=np.int64(-1), **ia, **tr)
        print("micro", name, int(tr["n_steps"]), float(tr["reward"]), flush=True)

    # test-set instances in RL mode (pkl durations are U(0,5), requirement 1..5): 6 hashed traces
    ts_hash = {}
    for i in range(6):
        env = load_testset_env(i)
        se = env_seed(2000, i)
        tr = rollout(env, se, POLICIES["random" if i % 2 == 0 else "nearest"], quirks=quirks)
        ts_hash[str(i)] = dict(policy="random" if i % 2 == 0 else "nearest", seed_e=str(se), sha256=digest(tr),
                               n_steps=int(tr["n_steps"]), reward=float(tr["reward"]))
    with open(f"{OUT}/testset_rl_hashes.json", "w") as f:
        json.dump(ts_hash, f, indent=1)

    # multi-episode determinism: d continues across auto-reset (3 episodes, same instance)
    env = TaskEnv((20, 20), (50, 50), 1, 5, seed=7)
    se = env_seed(3000, 0)
    d0, eps = 0, []
    for ep in range(3):
        env.reset()
        env.clear_decisions()
        tr = rollout(env, se, POLICIES["random"], d0=d0, record=False)
        d0 += int(tr["n_steps"])
        eps.append(dict(n_steps=int(tr["n_steps"]), reward=float(tr["reward"]),
                        n_finished=int(tr["finished"].sum()), metrics=[float(x) for x in tr["metrics"]]))
    manifest["multi_episode"] = dict(inst_seed=7, seed_e=str(se), episodes=eps)

    # instance generation with tuple ranges (sizes drawn from the seeded stream first, env/task_env.py:58-65)
    rng_inst = {}
    for sd in range(6):
        env = TaskEnv((10, 20), (20, 50), 1, 5, seed=sd)   # AGENTS_RANGE / TASKS_RANGE of parameters.py:15-16
        ia = instance_arrays(env)
        rng_inst[str(sd)] = dict(A=int(ia["A"]), T=int(ia["T"]), depot=ia["depot"].tolist(), task_xy0=ia["task_xy"][0].tolist(),
                                 task_xy_last=ia["task_xy"][-1].tolist(), req=ia["req"].tolist())
    with open(f"{OUT}/instances_ranges.json", "w") as f:
        json.dump(rng_inst, f)

    manifest["quirks_in_traces"] = quirks
    with open(f"{OUT}/manifest.json", "w") as f:
        json.dump(manifest, f, indent=1)
    print(json.dumps(manifest, indent=1))


def _fma(a, b, c):
    """Exact fused multiply-add via integer arithmetic on the fp64 mantissas (Python < 3.13)."""
    from fractions import Fraction

    r = Fraction(a) * Fraction(b) + Fraction(c)
    return float(r)  # Fraction -> float is correctly rounded


if __name__ == "__main__":
    main()
