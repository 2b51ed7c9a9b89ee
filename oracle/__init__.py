"""ctypes binding of the CPU oracle -- TEST INFRASTRUCTURE, NOT THE PRODUCT.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package (see oracle/dcmrta_oracle.h).  Nothing under dcmrta_amd/ does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

POLICY_RANDOM, POLICY_INJECTED, POLICY_FIRST, POLICY_NEAREST, POLICY_ANY = 0, 1, 2, 3, 4
METRIC_NAMES = ("success_rate", "makespan", "time_cost", "waiting_time", "travel_dist", "efficiency")


def build(force=False):
    """Compile oracle/liboracle.so with gcc (oracle/Makefile)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "dcmrta_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return so


class _Summary(C.Structure):
    _fields_ = [("reward", C.c_double), ("makespan", C.c_double), ("metrics", C.c_double * 6),
                ("truncated", C.c_int32), ("n_finished", C.c_int32)]


def lib():
    global _LIB
    if _LIB is None:
        so = os.environ.get("DCM_ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")   # (DCM_ORACLE_LIB: e.g. liboracle_asan.so)
        if not os.path.exists(so):
            so = build()
        L = C.CDLL(so)
        vp, i32, i64, u64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_double
        L.orc_create.restype = vp
        L.orc_create.argtypes = [C.c_int, C.c_int]
        L.orc_destroy.argtypes = [vp]
        L.orc_load_instance.argtypes = [vp, vp, vp, vp, vp]
        L.orc_clear_decisions.argtypes = [vp]
        L.orc_set_params.argtypes = [vp, dbl, dbl]
        for f in ("orc_mix64",):
            getattr(L, f).restype = u64
            getattr(L, f).argtypes = [u64]
        L.orc_env_seed.restype = u64
        L.orc_env_seed.argtypes = [u64, u64]
        L.orc_draw.restype = u64
        L.orc_draw.argtypes = [u64, u64, u64]
        L.orc_next_decision.restype = C.c_int
        L.orc_next_decision.argtypes = [vp, vp, vp]
        L.orc_get_unique_group.restype = C.c_int
        L.orc_get_unique_group.argtypes = [vp, vp, C.c_int, vp]
        L.orc_task_update.argtypes = [vp]
        L.orc_agent_update.argtypes = [vp]
        L.orc_agent_step.argtypes = [vp, C.c_int, C.c_int]
        L.orc_mask.argtypes = [vp, vp]
        L.orc_agent_status.argtypes = [vp, C.c_int, vp]
        L.orc_task_status.argtypes = [vp, C.c_int, vp]
        L.orc_check_finished.restype = C.c_int
        L.orc_check_finished.argtypes = [vp]
        L.orc_get_now.restype = dbl
        L.orc_get_now.argtypes = [vp]
        L.orc_set_now.argtypes = [vp, dbl]
        L.orc_rollout.restype = i64
        L.orc_rollout.argtypes = [vp, u64, u64, C.c_int, i64] + [vp] * 12
        L.orc_summary_get.argtypes = [vp, C.POINTER(_Summary)]
        L.orc_max_members_seen.restype = C.c_int
        L.orc_max_members_seen.argtypes = [vp]
        L.orc_final_tasks.argtypes = [vp] * 8
        L.orc_final_agents.argtypes = [vp] * 5
        L.orc_get_route.restype = C.c_int
        L.orc_get_route.argtypes = [vp, C.c_int, vp, vp, C.c_int]
        for f in ("orc_get_members", "orc_get_abandoned"):
            getattr(L, f).restype = C.c_int
            getattr(L, f).argtypes = [vp, C.c_int, vp, C.c_int]
        L.orc_pre_set_route.argtypes = [vp, C.c_int, vp, C.c_int]
        L.orc_set_visibility.restype = C.c_int
        L.orc_set_visibility.argtypes = [vp] + [C.c_int] * 4
        L.orc_execute_by_route.restype = C.c_int
        L.orc_execute_by_route.argtypes = [vp, C.c_int]
        L.orc_finish_episode.argtypes = [vp]
        L.orc_pairwise_sum.restype = dbl
        L.orc_pairwise_sum.argtypes = [vp, i64]
        L.orc_batch_rollout.restype = i64
        L.orc_batch_rollout.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp]
        L.orc_batch_rollout_ex.restype = i64
        L.orc_batch_rollout_ex.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp]
        L.orc_batch_replay.restype = i64
        L.orc_batch_replay.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp, C.c_int,
                                       vp, vp, vp, vp, vp]
        _LIB = L
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def mix64(z):
    return int(lib().orc_mix64(C.c_uint64(z & (2**64 - 1))))


def env_seed(base, e):
    return int(lib().orc_env_seed(C.c_uint64(base & (2**64 - 1)), C.c_uint64(e)))


def draw(seed_e, d, slot):
    return int(lib().orc_draw(C.c_uint64(seed_e), C.c_uint64(d), C.c_uint64(slot)))


def pairwise_sum(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return float(lib().orc_pairwise_sum(_p(a), a.size))


class OracleEnv:
    """One sequential env (the reference's TaskEnv restated in C)."""

    def __init__(self, A, T, max_waiting_time=10.0, max_time=100.0):
        self.A, self.T = int(A), int(T)
        self._h = C.c_void_p(lib().orc_create(self.A, self.T))
        lib().orc_set_params(self._h, float(max_waiting_time), float(max_time))

    def __del__(self):
        if getattr(self, "_h", None) and _LIB is not None:   # (module globals may already be gone at interpreter exit)
            _LIB.orc_destroy(self._h)
            self._h = None

    def load(self, depot, task_xy, req, dur):
        self._keep = (np.ascontiguousarray(depot, np.float64), np.ascontiguousarray(task_xy, np.float64),
                      np.ascontiguousarray(req, np.int32), np.ascontiguousarray(dur, np.float64))
        assert self._keep[1].shape == (self.T, 2) and self._keep[2].shape == (self.T,)
        lib().orc_load_instance(self._h, *[_p(x) for x in self._keep])
        return self

    def clear_decisions(self):
        lib().orc_clear_decisions(self._h)

    def set_params(self, max_waiting_time, max_time):
        lib().orc_set_params(self._h, float(max_waiting_time), float(max_time))

    @property
    def now(self):
        return float(lib().orc_get_now(self._h))

    @now.setter
    def now(self, v):
        lib().orc_set_now(self._h, float(v))

    # step-wise surface ---------------------------------------------------------------------
    def next_decision(self):
        ids = np.zeros(self.A, np.int32)
        t = C.c_double()
        n = lib().orc_next_decision(self._h, _p(ids), C.byref(t))
        return ids[:n].copy(), t.value

    def get_unique_group(self, ids):
        ids = np.ascontiguousarray(ids, np.int32)
        gof = np.zeros(max(len(ids), 1), np.int32)
        ng = lib().orc_get_unique_group(self._h, _p(ids), len(ids), _p(gof))
        return [[int(a) for a, g in zip(ids, gof) if g == k] for k in range(ng)]

    def task_update(self):
        lib().orc_task_update(self._h)

    def agent_update(self):
        lib().orc_agent_update(self._h)

    def agent_step(self, agent, action):
        lib().orc_agent_step(self._h, int(agent), int(action))

    def mask(self):
        m = np.zeros(self.T + 1, np.uint8)
        lib().orc_mask(self._h, _p(m))
        return m

    def agent_status(self, leader):
        o = np.zeros((self.A, 6), np.float32)
        lib().orc_agent_status(self._h, int(leader), _p(o))
        return o

    def task_status(self, leader):
        o = np.zeros((self.T + 1, 5), np.float32)
        lib().orc_task_status(self._h, int(leader), _p(o))
        return o

    def check_finished(self):
        return bool(lib().orc_check_finished(self._h))

    # whole episodes ------------------------------------------------------------------------
    def rollout(self, seed_e, d0=0, policy=POLICY_RANDOM, cap_steps=4096, record=True,
                inj_leader=None, inj_action=None, inj_nfol=None, inj_followers=None, allow_cap=False):
        A, T = self.A, self.T
        n = int(cap_steps)
        rec = {}
        if record:
            rec = dict(leader=np.zeros(n, np.int32), action=np.zeros(n, np.int32), nfol=np.zeros(n, np.int32),
                       followers=np.full((n, A), -1, np.int16), now=np.zeros(n, np.float64),
                       mask=np.zeros((n, T + 1), np.uint8), agents_obs=np.zeros((n, A, 6), np.float32),
                       tasks_obs=np.zeros((n, T + 1, 5), np.float32))
        inj = [None if x is None else np.ascontiguousarray(x, dt) for x, dt in
               ((inj_leader, np.int32), (inj_action, np.int32), (inj_nfol, np.int32), (inj_followers, np.int16))]
        order = ("leader", "action", "nfol", "followers", "now", "mask", "agents_obs", "tasks_obs")
        steps = lib().orc_rollout(self._h, C.c_uint64(seed_e), C.c_uint64(d0), int(policy), n,
                                  *[_p(x) for x in inj], *[_p(rec.get(k)) for k in order])
        if steps < 0:
            if not allow_cap:
                raise RuntimeError("oracle rollout exceeded cap_steps")
            steps = n        # the first cap_steps decisions were recorded; the episode is NOT over (no terminal results)
        out = {k: v[:steps] for k, v in rec.items()}
        out["n_steps"] = int(steps)
        out.update(self.final())
        return out

    def final(self):
        A, T = self.A, self.T
        s = _Summary()
        lib().orc_summary_get(self._h, C.byref(s))
        fin, fea = np.zeros(T, np.uint8), np.zeros(T, np.uint8)
        ts, tf, tw = np.zeros(T), np.zeros(T), np.zeros(T)
        nm, na = np.zeros(T, np.int32), np.zeros(T, np.int32)
        lib().orc_final_tasks(self._h, _p(fin), _p(fea), _p(ts), _p(tf), _p(tw), _p(nm), _p(na))
        aw, td = np.zeros(A), np.zeros(A)
        ret, rl = np.zeros(A, np.uint8), np.zeros(A, np.int32)
        lib().orc_final_agents(self._h, _p(aw), _p(td), _p(ret), _p(rl))
        return dict(reward=s.reward, makespan=s.makespan, metrics=np.array(list(s.metrics)), truncated=int(s.truncated),
                    finished=fin, feasible=fea, time_start=ts, time_finish=tf, task_wait=tw, n_members=nm,
                    n_abandoned=na, agent_wait=aw, travel_dist=td, returned=ret, route_len=rl,
                    max_members_seen=int(lib().orc_max_members_seen(self._h)))

    def route(self, agent, cap=4096):
        """(route, arrival_time) lists of the agent (env/task_env.py:95-96)."""
        t, a = np.zeros(cap, np.int32), np.zeros(cap, np.float64)
        n = lib().orc_get_route(self._h, int(agent), _p(t), _p(a), cap)
        return t[:n].copy(), a[:n].copy()

    def members(self, task, cap=256):
        """task['members'] in list order (env/task_env.py:78)."""
        out = np.zeros(cap, np.int32)
        n = lib().orc_get_members(self._h, int(task), _p(out), cap)
        return out[:n].copy()

    def abandoned(self, task, cap=4096):
        """task['abandoned_agent'] in append order (env/task_env.py:89)."""
        out = np.zeros(cap, np.int32)
        n = lib().orc_get_abandoned(self._h, int(task), _p(out), cap)
        return out[:n].copy()

    # route replay --------------------------------------------------------------------------
    def pre_set_route(self, actions, agent):
        a = np.ascontiguousarray(actions, np.int32)
        lib().orc_pre_set_route(self._h, int(agent), _p(a), len(a))

    def set_visibility(self, initial=20, batch=20, period=10, cap=100):
        """The four constants of the dynamic-arrival schedule (env/task_env.py:567, :221); defaults = the reference."""
        if lib().orc_set_visibility(self._h, int(initial), int(batch), int(period), int(cap)) != 0:
            raise ValueError("need initial >= 0, batch >= 1, period >= 1, cap >= initial")
        return self

    def execute_by_route(self, reactive=False):
        rc = lib().orc_execute_by_route(self._h, int(bool(reactive)))
        if rc == -2:
            raise TypeError("'NoneType' object is not subscriptable (reference env/task_env.py:220)")
        lib().orc_finish_episode(self._h)
        return self.final()


def batch_rollout(depot, task_xy, req, dur, seeds, A, episodes=1, threads=1):
    """CPU baseline: B envs, random policy, `threads` pthreads. Returns (total_steps, reward[B], steps[B], metrics[B,6])."""
    depot = np.ascontiguousarray(depot, np.float64)
    task_xy = np.ascontiguousarray(task_xy, np.float64)
    req = np.ascontiguousarray(req, np.int32)
    dur = np.ascontiguousarray(dur, np.float64)
    seeds = np.ascontiguousarray(seeds, np.uint64)
    B, T = req.shape
    reward, steps, metrics = np.zeros(B), np.zeros(B, np.int64), np.zeros((B, 6))
    total = lib().orc_batch_rollout(B, int(A), T, _p(depot), _p(task_xy), _p(req), _p(dur), _p(seeds), int(episodes),
                                    int(threads), _p(reward), _p(steps), _p(metrics))
    return int(total), reward, steps, metrics


def batch_rollout_full(depot, task_xy, req, dur, seeds, A, episodes=1, threads=1):
    """batch_rollout plus every episode's return and the last episode's finished-task count: dict(total, reward[B], steps[B],
    metrics[B,6], returns[B,episodes], n_finished[B]) -- what the full-batch parity checks compare the HIP path with."""
    depot = np.ascontiguousarray(depot, np.float64)
    task_xy = np.ascontiguousarray(task_xy, np.float64)
    req = np.ascontiguousarray(req, np.int32)
    dur = np.ascontiguousarray(dur, np.float64)
    seeds = np.ascontiguousarray(seeds, np.uint64)
    B, T = req.shape
    reward, steps, metrics = np.zeros(B), np.zeros(B, np.int64), np.zeros((B, 6))
    returns, nfin = np.zeros((B, int(episodes))), np.zeros(B, np.int32)
    total = lib().orc_batch_rollout_ex(B, int(A), T, _p(depot), _p(task_xy), _p(req), _p(dur), _p(seeds), int(episodes),
                                       int(threads), _p(reward), _p(steps), _p(metrics), _p(returns), _p(nfin))
    return dict(total=int(total), reward=reward, steps=steps, metrics=metrics, returns=returns, n_finished=nfin)


def batch_replay(depot, task_xy, req, dur, routes, route_len, reactive=False, visibility=None, threads=1):
    """execute_by_route for a batch (routes int32[B,A,cap], route_len int32[B,A], -1 = None): dict(total, reward[B], steps[B]
    = agent_step calls, metrics[B,6], n_finished[B], status[B]: 0 ok / 1 guard-truncated / 2 TypeError)."""
    depot = np.ascontiguousarray(depot, np.float64)
    task_xy = np.ascontiguousarray(task_xy, np.float64)
    req = np.ascontiguousarray(req, np.int32)
    dur = np.ascontiguousarray(dur, np.float64)
    routes = np.ascontiguousarray(routes, np.int32)
    route_len = np.ascontiguousarray(route_len, np.int32)
    B, T = req.shape
    A = route_len.shape[1]
    assert routes.shape[:2] == (B, A)
    vis = None if visibility is None else np.ascontiguousarray(visibility, np.int32)
    reward, steps, metrics = np.zeros(B), np.zeros(B, np.int64), np.zeros((B, 6))
    nfin, status = np.zeros(B, np.int32), np.zeros(B, np.int32)
    total = lib().orc_batch_replay(B, A, T, _p(depot), _p(task_xy), _p(req), _p(dur), _p(routes), _p(route_len),
                                   int(routes.shape[2]), int(bool(reactive)), _p(vis), int(threads), _p(reward), _p(steps),
                                   _p(metrics), _p(nfin), _p(status))
    return dict(total=int(total), reward=reward, steps=steps, metrics=metrics, n_finished=nfin, status=status)
