"""Shared helpers of the parity tests (test infrastructure)."""
import glob
import hashlib
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DIGEST_KEYS = ("leader", "action", "nfol", "followers", "now", "mask", "agents_obs", "tasks_obs", "metrics", "finished",
               "time_start", "travel_dist", "agent_wait", "task_wait")
FINAL_EXACT = ("finished", "feasible", "time_start", "time_finish", "task_wait", "n_members", "n_abandoned",
               "travel_dist", "returned", "agent_wait")


def full_traces():
    """Full golden step traces: trace_<A>A<T>T_<policy>_s<seed>.npz and the hand-built tie scenarios micro_*.npz."""
    return sorted(glob.glob(os.path.join(GOLDEN, "trace_*.npz"))) + sorted(glob.glob(os.path.join(GOLDEN, "micro_*.npz")))


def trace_policy(path):
    base = os.path.basename(path)
    return "first" if base.startswith("micro_") else base.split("_")[2]


def load_trace(path):
    z = np.load(path)
    return {k: z[k] for k in z.files}


def digest(tr):
    h = hashlib.sha256()
    for k in DIGEST_KEYS:
        h.update(np.ascontiguousarray(tr[k]).tobytes())
    return h.hexdigest()


def trace_hashes():
    with open(os.path.join(GOLDEN, "trace_hashes.json")) as f:
        return json.load(f)


def host_random_action(mask_row, seed_e, d):
    """Uniform-random valid action of the choice protocol (slot 1), host mirror."""
    from dcmrta_amd.choice import below, draw
    valid = np.flatnonzero(mask_row == 0)
    return int(valid[below(draw(seed_e, d, 1), len(valid))])


def run_lockstep(env, seeds, policy, inject=None, max_iters=100000):
    """Drive a BatchedTaskEnv step by step (fused observe) and record per-env traces.

    policy(b, i, mask_row, leader) -> action for env b at its i-th decision.
    inject: optional dict of per-env arrays leader[b][i], nfol[b][i], followers[b][i,:] replayed as injected choices.
    Returns list of dicts with per-step arrays (same keys as the golden traces) per env.
    """
    import torch
    B, A, T = env.B, env.A, env.T
    if inject is not None:
        lead0 = np.array([inject["leader"][b][0] if len(inject["leader"][b]) else -1 for b in range(B)], np.int32)
        env.reset(seeds, observe=False)
        obs = env.observe(leader=lead0)
    else:
        obs = env.reset(seeds)
    rec = [dict(leader=[], action=[], now=[], mask=[], agents_obs=[], tasks_obs=[]) for _ in range(B)]
    count = np.zeros(B, np.int64)
    for _ in range(max_iters):
        active = obs.active.cpu().numpy()
        if not active.any():
            break
        ag = obs.agents.cpu().numpy()
        tk = obs.tasks.cpu().numpy()
        mk = obs.mask.cpu().numpy().astype(np.uint8)
        ld = obs.leader.cpu().numpy()
        now = env.status()["now"].cpu().numpy()
        actions = np.zeros(B, np.int32)
        nfol = np.full(B, -1, np.int32)
        fol = np.full((B, 4), -1, np.int16)
        nxt_leader = np.full(B, -1, np.int32)
        for b in range(B):
            if not active[b]:
                continue
            i = int(count[b])
            a = policy(b, i, mk[b], int(ld[b]))
            actions[b] = a
            r = rec[b]
            r["leader"].append(int(ld[b])); r["action"].append(a); r["now"].append(float(now[b]))
            r["mask"].append(mk[b].copy()); r["agents_obs"].append(ag[b].copy()); r["tasks_obs"].append(tk[b].copy())
            if inject is not None:
                nfol[b] = inject["nfol"][b][i]
                f = inject["followers"][b][i][:4]
                fol[b, :len(f)] = f
                if a == 0:
                    nfol[b] = -1  # depot: the whole group leaves, nothing to inject
                if i + 1 < len(inject["leader"][b]):
                    nxt_leader[b] = inject["leader"][b][i + 1]
            count[b] += 1
        if inject is not None:
            env.step(actions, n_followers=nfol, followers=fol, observe=False)
            obs = env.observe(leader=nxt_leader)
        else:
            obs = env.step(actions)
    else:
        raise RuntimeError("run_lockstep did not terminate")
    out = []
    for b in range(B):
        r = rec[b]
        n = len(r["leader"])
        out.append(dict(
            n_steps=n, leader=np.array(r["leader"], np.int32), action=np.array(r["action"], np.int32),
            now=np.array(r["now"], np.float64),
            mask=np.stack(r["mask"]) if n else np.zeros((0, T + 1), np.uint8),
            agents_obs=np.stack(r["agents_obs"]) if n else np.zeros((0, A, 6), np.float32),
            tasks_obs=np.stack(r["tasks_obs"]) if n else np.zeros((0, T + 1, 5), np.float32)))
    return out


def gpu_final(env):
    """Terminal arrays of every env in the golden/oracle naming."""
    ts = {k: v.cpu().numpy() for k, v in env.tasks_state().items()}
    ag = {k: v.cpu().numpy() for k, v in env.agents_state().items()}
    sm = env.summary().cpu().numpy()
    st = {k: v.cpu().numpy() for k, v in env.status().items()}
    out = []
    for b in range(env.B):
        out.append(dict(
            reward=sm[b, 0], n_finished=sm[b, 1], metrics=sm[b, 2:8].copy(), makespan=sm[b, 3], flags=int(st["flags"][b]),
            finished=ts["finished"][b], feasible=ts["feasible"][b], time_start=ts["time_start"][b],
            time_finish=ts["time_finish"][b], task_wait=ts["sum_waiting_time"][b], n_members=ts["n_members"][b],
            n_abandoned=ts["n_abandoned"][b], agent_wait=ag["sum_waiting_time"][b], travel_dist=ag["travel_dist"][b],
            returned=ag["returned"][b]))
    return out


def assert_final_matches(got, ref, name=""):
    """Every terminal quantity bit-exact: flags, counts, times, per-task / per-agent waiting sums, the six perf metrics."""
    for k in FINAL_EXACT:
        a, b = np.asarray(got[k]), np.asarray(ref[k])
        assert a.shape == b.shape and np.array_equal(a.astype(b.dtype), b), f"{name}: {k} differs"
    m, r = np.asarray(got["metrics"]), np.asarray(ref["metrics"])
    for i in range(6):  # success_rate, makespan, time_cost, waiting_time, travel_dist, efficiency
        assert m[i] == r[i], f"{name}: metric {i} {m[i]!r} != {r[i]!r}"
    assert float(got["reward"]) == float(ref["reward"]), f"{name}: reward"
