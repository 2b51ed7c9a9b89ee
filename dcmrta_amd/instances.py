"""Problem instances for the batched env (host side, numpy only).

generate_instance restates the draw order of the reference generator
(env/task_env.py:57-114) for fixed sizes: with a seeded numpy Generator the order is
depot random((1,2)) :67 -> cost random((A,1)) :68 (unused, but it advances the stream)
-> task xy random((T,2)) :69 -> requirements integers(1, max_coalition_size+1, T) :71;
durations are the constant max_duration :70.  (When agents_range/tasks_range are tuples
the reference first draws the sizes :59,:63; the batched env needs one (A,T) per batch,
exactly like one driver round does, driver.py:114-117.)
"""
import json
import os

import numpy as np


def generate_instance(A, T, seed, max_coalition_size=5, max_duration=5.0):
    rng = np.random.default_rng(seed)
    depot = rng.random((1, 2))[0]
    rng.random((A, 1))  # agents' cost: drawn, never used (env/task_env.py:68,97)
    task_xy = rng.random((T, 2))
    req = rng.integers(1, max_coalition_size + 1, T).astype(np.int32)
    dur = np.full(T, float(max_duration), dtype=np.float64)
    return dict(depot=depot, task_xy=task_xy, req=req, dur=dur)


def generate_instance_ranges(agents_range, tasks_range, seed, max_coalition_size=5, max_duration=5.0):
    """TaskEnv(agents_range, tasks_range, ...) with tuple ranges: the sizes are drawn first, tasks then agents
    (env/task_env.py:58-65), from the same seeded stream.  Returns (A, instance dict)."""
    rng = np.random.default_rng(seed)
    T = int(rng.integers(tasks_range[0], tasks_range[1] + 1)) if isinstance(tasks_range, tuple) else int(tasks_range)
    A = int(rng.integers(agents_range[0], agents_range[1] + 1)) if isinstance(agents_range, tuple) else int(agents_range)
    depot = rng.random((1, 2))[0]
    rng.random((A, 1))
    task_xy = rng.random((T, 2))
    req = rng.integers(1, max_coalition_size + 1, T).astype(np.int32)
    return A, dict(depot=depot, task_xy=task_xy, req=req, dur=np.full(T, float(max_duration)))


def generate_batch(B, A, T, base_seed=0, first=0, max_coalition_size=5, max_duration=5.0):
    """Instances base_seed+first .. base_seed+first+B-1 stacked as depot[B,2], task_xy[B,T,2], req[B,T], dur[B,T]."""
    depot = np.empty((B, 2), np.float64)
    task_xy = np.empty((B, T, 2), np.float64)
    req = np.empty((B, T), np.int32)
    for b in range(B):
        rng = np.random.default_rng(base_seed + first + b)
        depot[b] = rng.random((1, 2))[0]
        rng.random((A, 1))
        task_xy[b] = rng.random((T, 2))
        req[b] = rng.integers(1, max_coalition_size + 1, T)
    dur = np.full((B, T), float(max_duration), dtype=np.float64)
    return dict(depot=depot, task_xy=task_xy, req=req, dur=dur)


def generate_batch_ranges(seeds, agents_range, tasks_range, max_coalition_size=5, max_duration=5.0):
    """Ragged batch: instance b = TaskEnv(agents_range, tasks_range, seed=seeds[b]) (runner.py:45-49 with the tuple
    ranges of parameters.py:15-16).  Arrays are padded to the range maxima (task rows beyond n_tasks[b]: xy 0, req 1,
    dur 0 -- ignored by dcm_load_instances_ragged); returns the load_instances keyword dict incl. n_agents / n_tasks."""
    seeds = list(seeds)
    B = len(seeds)
    A = int(agents_range[1]) if isinstance(agents_range, tuple) else int(agents_range)
    T = int(tasks_range[1]) if isinstance(tasks_range, tuple) else int(tasks_range)
    out = dict(depot=np.zeros((B, 2)), task_xy=np.zeros((B, T, 2)), req=np.ones((B, T), np.int32), dur=np.zeros((B, T)),
               n_agents=np.zeros(B, np.int32), n_tasks=np.zeros(B, np.int32))
    for b, s in enumerate(seeds):
        a, inst = generate_instance_ranges(agents_range, tasks_range, int(s), max_coalition_size, max_duration)
        t = inst["req"].shape[0]
        out["depot"][b] = inst["depot"]
        out["task_xy"][b, :t] = inst["task_xy"]
        out["req"][b, :t] = inst["req"]
        out["dur"][b, :t] = inst["dur"]
        out["n_agents"][b], out["n_tasks"][b] = a, t
    return out


def instance_from_dicts(task_dic, agent_dic, depot):
    """(A, instance dict) from the reference's own containers (env/task_env.py:76-113): what `env.reset(test_env)` of
    RL_test.py:36-42 / baselines/CTAS-D.py:60-66 receives after unpickling a test-set env."""
    T = len(task_dic)
    first = lambda x: np.asarray(x, dtype=np.float64).reshape(-1)[0]
    return len(agent_dic), dict(
        depot=np.asarray(depot["location"], dtype=np.float64).reshape(-1)[:2].copy(),
        task_xy=np.stack([np.asarray(task_dic[i]["location"], dtype=np.float64).reshape(-1)[:2] for i in range(T)]),
        req=np.array([int(first(task_dic[i]["requirements"])) for i in range(T)], np.int32),
        dur=np.array([first(task_dic[i]["time"]) for i in range(T)], np.float64))


def batch_from_dicts(test_envs):
    """Stack several (task_dic, agent_dic, depot) triples of equal sizes into the load_instances keyword dict; returns (A, dict)."""
    parts = [instance_from_dicts(*te) for te in test_envs]
    A = parts[0][0]
    if any(a != A or p["req"].shape != parts[0][1]["req"].shape for a, p in parts):
        raise ValueError("all instances of a uniform batch need the same agent and task counts (use n_agents / n_tasks for ragged batches)")
    return A, {k: np.stack([p[k] for _, p in parts]) for k in ("depot", "task_xy", "req", "dur")}


def load_instances_npz(path):
    """Fixture format of tests/golden/instances_20A50T.npz (depot[N,2], task_xy[N,T,2], req[N,T], dur[N,T], A)."""
    z = np.load(path)
    return dict(depot=z["depot"], task_xy=z["task_xy"], req=z["req"].astype(np.int32), dur=z["dur"]), int(z["A"])


def load_routes_json(path):
    """Preset routes {instance: [[0, k1, k2, ..., 0], ...]} in CTAS-D node numbering (baselines/CTAS-D.py:10-33)."""
    with open(path) as f:
        return {int(k): v for k, v in json.load(f).items()}


def synthetic_routes(req, A, max_task=None):
    """Preset routes for the route-replay benchmark (the reference ships routes for 20A/50T only):
    every task t is visited by req[t] agents ((7t + j) mod A); each agent visits its tasks in ascending id, then the depot.
    max_task: only the first max_task tasks are routed (the reference's visibility cap hides the rest)."""
    T = len(req)
    r = [[] for _ in range(A)]
    for t in range(T if max_task is None else min(T, max_task)):
        for j in range(int(req[t])):
            r[(7 * t + j) % A].append(t + 1)
    return [x + [0] for x in r]


def synthetic_route_arrays(req, A, max_task=None):
    """synthetic_routes for a whole batch as arrays: req int[B,T] -> (routes int32[B,A,cap], route_len int32[B,A]) in the
    format of dcm_load_routes (0-padded; every agent's list ends with the depot action 0)."""
    req = np.asarray(req)
    B, T = req.shape
    Tm = T if max_task is None else min(T, int(max_task))
    t_idx = np.repeat(np.arange(Tm), 5)                      # candidate visits (t, j), j < 5 >= any requirement
    j_idx = np.tile(np.arange(5), Tm)
    agent = (7 * t_idx + j_idx) % A
    lens = np.zeros((B, A), np.int64)
    per_env = []
    for b in range(B):
        keep = j_idx < req[b, t_idx]
        a, t = agent[keep], t_idx[keep]
        order = np.argsort(a, kind="stable")                 # per agent, tasks stay in ascending id
        a, t = a[order], t[order]
        cnt = np.bincount(a, minlength=A)
        lens[b] = cnt
        per_env.append((a, t, cnt))
    cap = int(lens.max()) + 1
    routes = np.zeros((B, A, cap), np.int32)
    for b, (a, t, cnt) in enumerate(per_env):
        start = np.concatenate([[0], np.cumsum(cnt)[:-1]])
        pos = np.arange(len(a)) - start[a]
        routes[b, a, pos] = t + 1
    return routes, (lens + 1).astype(np.int32)               # + the trailing depot action
