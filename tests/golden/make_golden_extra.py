#!/usr/bin/env python3
"""Further reference-generated golden digests (run in the build container, where /root/reference is importable):

    python tests/golden/make_golden_extra.py   ->  tests/golden/trace_hashes_extra.json

Same harness as make_golden.py (the reference TaskEnv driven through the loop of worker.py:45-87 with the keyed
choice protocol injected), on the axes the first set does not vary:

  ranges      TaskEnv((10,20),(20,50),1,5,seed=s): per-seed sizes drawn by the reference itself (env/task_env.py:58-65)
  mwt         env.max_waiting_time in {3, 25}   (RL_test.py / baselines set it after construction, like load_testset_env)
  max_time    loop bound of worker.py:45 in {30, 250}
  coalition   max_coalition_size 3 (requirements 1..3) and max_duration 2 / 0
  wide        more agents than tasks
  coincident  tasks sharing coordinates, a task on the depot

and replay_random.json: the reference's execute_by_route on random preset routes.

Only numbers are stored (sizes, seeds, digests, rewards); no reference source.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from make_golden import TaskEnv  # noqa: E402


def run(env, seed_e, policy, max_time=100):
    old = mg.MAX_TIME
    mg.MAX_TIME = max_time
    try:
        return mg.rollout(env, seed_e, mg.POLICIES[policy])
    finally:
        mg.MAX_TIME = old


OBS_KEYS = ("leader", "action", "now", "mask", "agents_obs", "tasks_obs", "metrics", "finished", "time_start", "travel_dist",
            "agent_wait", "task_wait")


def digest_obs(tr):
    """Digest over what the lockstep API hands back at every decision (no follower lists): the GPU parity test
    recomputes it from the device outputs alone."""
    import hashlib
    h = hashlib.sha256()
    for k in OBS_KEYS:
        h.update(np.ascontiguousarray(tr[k]).tobytes())
    return h.hexdigest()


def entry(env, tr, seed_e, policy, **kw):
    ia = mg.instance_arrays(env)
    d = dict(A=int(env.agents_num), T=int(env.tasks_num), policy=policy, seed_e=str(seed_e), sha256=mg.digest(tr),
             sha256_obs=digest_obs(tr),
             n_steps=int(tr["n_steps"]), reward=float(tr["reward"]), n_finished=int(tr["finished"].sum()),
             req_sum=int(ia["req"].sum()), dur0=float(ia["dur"][0]))
    d.update(kw)
    return d


def random_routes(rng, A, T, req):
    """Random preset routes in the style of tools/sweep_replay.py: per task req +0..2 -0..1 visitors (sometimes too few,
    sometimes surplus), visiting order sorted or shuffled, 5 % of the agents keep pre_set_route None, 80 % end with 0."""
    r = [[] for _ in range(A)]
    for t in range(T):
        k = min(A, int(req[t]) + int(rng.integers(0, 3)) - int(rng.integers(0, 2)))
        for a in rng.choice(A, size=max(k, 0), replace=False):
            r[int(a)].append(t + 1)
    routes = []
    for a in range(A):
        if rng.random() < 0.05:
            routes.append(None)
            continue
        x = r[a]
        if rng.random() < 0.5:
            x = sorted(x)
        else:
            rng.shuffle(x)
        routes.append([int(v) for v in x] + ([0] if rng.random() < 0.8 else []))
    return routes


class _Timeout(Exception):
    pass


def random_replays():
    """execute_by_route (env/task_env.py:562-593) of the reference on random routes: known answers for the corners the
    CTAS-D routes never reach (surplus visitors released early, too few visitors, None routes, shuffled order)."""
    import contextlib
    import copy
    import io
    import signal

    def on_alarm(sig, frm):
        raise _Timeout()
    signal.signal(signal.SIGALRM, on_alarm)
    rng = np.random.default_rng(99)
    cases = []
    shapes = [(20, 50, "testset")] * 6 + [(20, 3), (33, 7), (8, 7), (70, 3), (13, 20), (5, 37), (40, 20), (20, 20),
                                          (10, 64), (64, 10), (3, 3), (2, 7)]
    for ci, shp in enumerate(shapes):
        for reactive in (False, True):
            if len(shp) == 3:
                A, T = 20, 50
                env = mg.load_testset_env(ci)
                src = dict(kind="testset", index=ci)
            else:
                A, T = shp
                env = TaskEnv((A, A), (T, T), 1, 5, seed=100 + ci)
                src = dict(kind="fixed", inst_seed=100 + ci)
            ia = mg.instance_arrays(env)
            routes = random_routes(rng, A, T, ia["req"])
            env.reactive_planning = reactive
            for a, r in enumerate(routes):
                if r is not None:
                    env.pre_set_route(copy.copy(r), a)
            case = dict(A=A, T=T, reactive=reactive, routes=routes, req_sum=int(ia["req"].sum()), **src)
            signal.alarm(120)
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    env.execute_by_route("./", "x", False)
                env.get_episode_reward(100)
                fa = mg.final_arrays(env)
                case["result"] = {k: np.asarray(fa[k]).tolist() for k in
                                  ("metrics", "finished", "time_start", "time_finish", "task_wait", "agent_wait", "travel_dist",
                                   "returned", "n_members", "route_len")}
                case["status"] = "ok"
            except TypeError:
                case["status"] = "type_error"          # env/task_env.py:220 with pre_set_route None
            except _Timeout:
                case["status"] = "no_termination"      # the reference loops forever (SURVEY a-16); not a known answer
            finally:
                signal.alarm(0)
            print("replay", ci, A, T, reactive, case["status"], case.get("result", {}).get("metrics", [None, None])[:2], flush=True)
            cases.append(case)
    with open(os.path.join(HERE, "replay_random.json"), "w") as f:
        json.dump(cases, f, separators=(",", ":"))


def trajectory_goldens():
    """generate_traj (env/task_env.py:375-418) of the reference after one harness episode: the sampled positions of every
    agent plus everything a pure re-implementation needs as input (routes, arrivals, final member lists, task times)."""
    out = {}
    for (A, T, pol, s) in ((5, 8, "random", 3), (10, 20, "nearest", 4), (6, 9, "random", 5)):
        env = TaskEnv((A, A), (T, T), 1, 5, seed=s)
        se = mg.env_seed(5600, s)
        tr = run(env, se, pol)
        env.generate_traj()
        ia = mg.instance_arrays(env)
        rl = max(len(a["route"]) for a in env.agent_dic.values())
        route = np.full((A, rl), -2, np.int32)
        arrival = np.zeros((A, rl))
        for a, ag in env.agent_dic.items():
            route[a, :len(ag["route"])] = ag["route"]
            arrival[a, :len(ag["arrival_time"])] = ag["arrival_time"]
        members = np.full((T, 5), -1, np.int16)
        for t, tk in env.task_dic.items():
            members[t, :len(tk["members"])] = tk["members"]
        traj = [np.array(ag["trajectory"], np.float64).reshape(-1, 3) for ag in env.agent_dic.values()]
        tl = np.array([len(x) for x in traj], np.int32)
        flat = np.concatenate(traj) if tl.sum() else np.zeros((0, 3))
        key = f"traj_{A}A{T}T_{pol}_s{s}"
        np.savez_compressed(os.path.join(HERE, key + ".npz"), seed_e=np.uint64(se), inst_seed=np.int64(s), depot=ia["depot"],
                            task_xy=ia["task_xy"], req=ia["req"], dur=ia["dur"], route=route, arrival=arrival, members=members,
                            feasible=tr["feasible"], time_start=tr["time_start"], time_finish=tr["time_finish"],
                            current_time=np.float64(env.current_time), action=tr["action"], traj=flat, traj_len=tl)
        print(key, int(tr["n_steps"]), tl.tolist(), flush=True)


def main():
    if "--only-traj" in sys.argv:
        trajectory_goldens()
        return
    trajectory_goldens()
    if "--skip-replays" not in sys.argv:
        random_replays()
    out = {}
    # ranges: the reference draws (T, A) itself
    for s in range(16):
        env = TaskEnv((10, 20), (20, 50), 1, 5, seed=s)
        se = mg.env_seed(5000, s)
        pol = "random" if s % 2 == 0 else "nearest"
        tr = run(env, se, pol)
        out[f"ranges_s{s}"] = entry(env, tr, se, pol, kind="ranges", inst_seed=s)
    # max_waiting_time
    for (A, T) in ((10, 20), (20, 50)):
        for mwt in (3, 25):
            for s in (0, 1):
                env = TaskEnv((A, A), (T, T), 1, 5, seed=s)
                env.max_waiting_time = mwt
                se = mg.env_seed(5100 + mwt, s)
                tr = run(env, se, "random")
                out[f"mwt{mwt}_{A}A{T}T_s{s}"] = entry(env, tr, se, "random", kind="fixed", inst_seed=s, max_waiting_time=mwt)
    # loop bound
    for (A, T) in ((10, 20), (20, 50)):
        for mt in (30, 250):
            env = TaskEnv((A, A), (T, T), 1, 5, seed=3)
            se = mg.env_seed(5200 + mt, 3)
            pol = "random" if mt == 30 else "nearest"
            tr = run(env, se, pol, max_time=mt)
            out[f"maxtime{mt}_{A}A{T}T"] = entry(env, tr, se, pol, kind="fixed", inst_seed=3, max_time=mt)
    # coalition size / duration
    for name, mcs, dur in (("coal3", 3, 5), ("dur2", 5, 2), ("dur0", 5, 0)):
        for s in (0, 1):
            env = TaskEnv((12, 12), (30, 30), 1, mcs, max_duration=dur, seed=s)
            se = mg.env_seed(5300, s)
            tr = run(env, se, "random")
            out[f"{name}_12A30T_s{s}"] = entry(env, tr, se, "random", kind="fixed", inst_seed=s, max_coalition_size=mcs,
                                               max_duration=dur)
    # more agents than tasks
    for (A, T) in ((30, 10), (64, 5)):
        env = TaskEnv((A, A), (T, T), 1, 5, seed=2)
        se = mg.env_seed(5400, A)
        tr = run(env, se, "random")
        out[f"wide_{A}A{T}T"] = entry(env, tr, se, "random", kind="fixed", inst_seed=2)
    # coincident locations: tasks sharing coordinates (agents deciding there at the same time form ONE group, since
    # get_unique_group looks at locations only, env/task_env.py:291-298) and a task on top of the depot.  The instance
    # arrays are stored with the digest (they are not what the seeded generator would produce).
    coincident = {}
    for s in range(6):
        A, T = (8, 12) if s < 3 else (20, 30)
        env = TaskEnv((A, A), (T, T), 1, 5, seed=200 + s)
        rng = np.random.default_rng(300 + s)
        for i in range(T):
            j = int(rng.integers(0, 4))
            if j < i and rng.random() < 0.6:
                env.task_dic[i]["location"] = env.task_dic[j]["location"].copy()      # equal value, distinct array
        env.task_dic[T - 1]["location"] = np.array(env.depot["location"], dtype=np.float64).copy()
        if s % 2:
            for i in range(T):
                env.task_dic[i]["requirements"] = np.array([1 + i % 2])
                env.task_dic[i]["status"] = np.array([1 + i % 2])
        env.clear_decisions()
        ia = mg.instance_arrays(env)
        se = mg.env_seed(5500, s)
        tr = run(env, se, "random")
        e = entry(env, tr, se, "random", kind="arrays")
        e.update(depot=ia["depot"].tolist(), task_xy=ia["task_xy"].tolist(), req=ia["req"].tolist(), dur=ia["dur"].tolist())
        out[f"coincident_s{s}"] = e
    for k, v in out.items():
        print(k, v["A"], v["T"], v["n_steps"], v["reward"], v["n_finished"], flush=True)
    with open(os.path.join(HERE, "trace_hashes_extra.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
