"""BatchedTaskEnv -- host-side face of the HIP env (B independent TaskEnv instances on one GPU).

Mirrors what worker.py:41-112 does around one reference TaskEnv, for a whole batch:

    env = BatchedTaskEnv(B, A, T, device="cuda:0")
    env.load_instances(depot, task_xy, req, dur)        # generate_env outputs  (env/task_env.py:57-114)
    obs = env.reset(seeds)                              # reset+clear_decisions (:116-140) -> first decision
    while obs.active.any():
        logp = policy(obs.tasks, obs.agents, obs.mask)  # attention.py:288-297 input contract
        obs = env.step(actions)                         # TaskEnv.step :326-342 + worker.py:74-85
    env.summary()                                       # reward + perf metrics, worker.py:87,103-108

torch is used for device memory and streams only; all simulation work happens in
libdcmrta_hip.so (hand-written gfx950 kernels) through the C ABI of include/dcmrta_env.h.
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from ._lib import DcmError, DcmParams, check

SUMMARY_COLUMNS = ("reward", "n_finished", "success_rate", "makespan", "time_cost", "waiting_time", "travel_dist",
                   "efficiency")


@dataclass
class Observation:
    """Inputs of the attention policy for every env (SURVEY §8b policy contract)."""
    agents: torch.Tensor  # f32[B,A,6]
    tasks: torch.Tensor   # f32[B,T+1,5]
    mask: torch.Tensor    # bool[B,T+1], True = forbidden
    leader: torch.Tensor  # i32[B], -1 when the env's episode is over
    active: torch.Tensor  # bool[B]


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


# hipStream_t of torch's current stream on a device as a plain int (the raw accessor skips the Stream object)
_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None) or (lambda i: torch.cuda.current_stream(i).cuda_stream)


class BatchedTaskEnv:
    def __init__(self, n_envs, n_agents, n_tasks, device="cuda:0", max_waiting_time=10.0, max_time=100.0,
                 individual_selection=False, auto_reset=False, auto_reset_episodes=0, strict_mask=False, member_cap=5):
        self._h = None
        self._lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise DcmError("BatchedTaskEnv runs on a HIP device only (device='cuda:N'); there is no CPU path")
        if not torch.cuda.is_available():
            raise DcmError("no HIP device visible to torch; BatchedTaskEnv has no CPU fallback")
        self.B, self.A, self.T = int(n_envs), int(n_agents), int(n_tasks)
        self.max_waiting_time, self.max_time = float(max_waiting_time), float(max_time)
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        # individual_selection: Worker.run_test_IS (worker.py:159-198) -- deciders are not grouped by location
        # auto_reset: step() restarts an env from its instance in the call that ends its episode (DCM_PARAM_AUTO_RESET): the
        # batch stays full; summary() holds each env's last finished episode, episodes() counts them
        # strict_mask: freeze an env whose host-supplied action is on a masked task (DCM_PARAM_STRICT_MASK) instead of
        # simulating it like the reference's TaskEnv.step does
        # member_cap: member slots per task -- 5 (COALITION_SIZE, parameters.py:17) or, for a mask-ignoring policy (worker.py:140) or
        # max_coalition_size > 5 (env/task_env.py:71), 16 (DCM_PARAM_WIDE_MEMBERS: larger records, runtime-size kernels, no replay)
        if not 1 <= int(member_cap) <= _lib.MAX_MEMBERS_WIDE:
            raise DcmError(f"member_cap must be in 1..{_lib.MAX_MEMBERS_WIDE}")
        self.member_cap = _lib.MAX_MEMBERS if int(member_cap) <= _lib.MAX_MEMBERS else _lib.MAX_MEMBERS_WIDE
        flags = (1 if individual_selection else 0) | (2 if auto_reset else 0) | (4 if strict_mask else 0) | \
                (_lib.PARAM_WIDE_MEMBERS if self.member_cap > _lib.MAX_MEMBERS else 0)
        # auto_reset_episodes: an env stops restarting after that many finished episodes (0 = never)
        p = DcmParams(self.B, self.A, self.T, idx, self.max_waiting_time, self.max_time, flags, int(auto_reset_episodes))
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(self._lib.dcm_create(C.byref(p), C.byref(h)))
        self._h = h
        B, A, T = self.B, self.A, self.T
        dev = self.device
        self._agents = torch.empty((B, A, 6), dtype=torch.float32, device=dev)
        self._tasks = torch.empty((B, T + 1, 5), dtype=torch.float32, device=dev)
        self._mask = torch.empty((B, T + 1), dtype=torch.uint8, device=dev)
        self._leader = torch.empty((B,), dtype=torch.int32, device=dev)
        self._active = torch.empty((B,), dtype=torch.uint8, device=dev)
        self._instances = None
        self.n_agents = self.n_tasks = None
        # the lockstep hot path (step() once per decision of a policy in the loop): everything that does not change between calls is
        # bound once -- the output pointers as plain ints, the Observation over the static buffers, the entry point
        self._out_ptrs = tuple(int(t.data_ptr()) for t in (self._agents, self._tasks, self._mask, self._leader, self._active))
        self._obs_static = Observation(self._agents, self._tasks, self._mask.view(torch.bool), self._leader, self._active.view(torch.bool))
        self._dcm_step = self._lib.dcm_step
        self._dev_index = idx
        # bumped whenever something a captured HIP graph of dcm_step has baked in changes (raggedness of the batch: the
        # per-env sizes pointer and the kernel instantiation; the route-log pointers): GraphedRollout re-captures
        self.graph_epoch = 0

    # ------------------------------------------------------------------ lifecycle
    def close(self):
        if getattr(self, "_h", None):
            self._lib.dcm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _dev(self, x, dtype):
        if isinstance(x, torch.Tensor):
            return x.to(device=self.device, dtype=dtype).contiguous()
        return torch.as_tensor(np.ascontiguousarray(x), dtype=dtype).to(self.device)

    # ------------------------------------------------------------------ instances / reset
    def load_instances(self, depot, task_xy, req, dur, n_agents=None, n_tasks=None):
        """depot[B,2] f64, task_xy[B,T,2] f64, req[B,T] int in 1..member_cap, dur[B,T] f64 (numpy or torch).

        n_agents[B] / n_tasks[B] (host ints, 1..A / 1..T) make the batch ragged: env e is an (n_agents[e], n_tasks[e])
        env (TaskEnv with tuple ranges, env/task_env.py:58-65), rows beyond its sizes in the arrays are ignored and its
        observation rows beyond them are padding (-1 features, mask True: attention.py:10-18, worker.py:253-261)."""
        B, T = self.B, self.T
        if (n_agents is None) != (n_tasks is None):
            raise DcmError("give both n_agents and n_tasks, or neither")
        req_np = (req.cpu().numpy() if isinstance(req, torch.Tensor) else np.asarray(req)).copy()
        if req_np.shape != (B, T):
            raise DcmError(f"req must be int[{B},{T}]")
        if n_tasks is not None:
            nt = np.ascontiguousarray(n_tasks, dtype=np.int32)
            na = np.ascontiguousarray(n_agents, dtype=np.int32)
            if nt.shape != (B,) or na.shape != (B,):
                raise DcmError(f"n_agents / n_tasks must be int[{B}]")
            req_np[np.arange(T)[None, :] >= nt[:, None]] = 1      # ignored rows: anything valid
        if req_np.min() < 1 or req_np.max() > self.member_cap:
            raise DcmError(f"req must be int[{B},{T}] with values in 1..{self.member_cap} (member_cap of this env)")
        d = self._dev(depot, torch.float64)
        xy = self._dev(task_xy, torch.float64)
        rq = self._dev(req, torch.int32)
        du = self._dev(dur, torch.float64)
        if d.shape != (B, 2) or xy.shape != (B, T, 2) or du.shape != (B, T):
            raise DcmError("instance arrays have the wrong shape")
        if (n_tasks is None) != (self.n_tasks is None):
            self.graph_epoch += 1
        with torch.cuda.device(self.device):
            if n_tasks is None:
                check(self._lib.dcm_load_instances(self._h, _ptr(d), _ptr(xy), _ptr(rq), _ptr(du), self._stream()))
                self.n_agents = self.n_tasks = None
            else:
                check(self._lib.dcm_load_instances_ragged(self._h, _ptr(d), _ptr(xy), _ptr(rq), _ptr(du),
                                                          na.ctypes.data_as(C.c_void_p), nt.ctypes.data_as(C.c_void_p),
                                                          self._stream()))
                self.n_agents, self.n_tasks = na.copy(), nt.copy()
        self._instances = (d, xy, rq, du)  # keep alive until the kernel ran
        return self

    def reset(self, seeds, observe=True):
        """seeds: uint64[B] per-env choice-protocol seeds (numpy/torch) or an int base seed."""
        if isinstance(seeds, (int, np.integer)):
            from .choice import env_seeds
            seeds = env_seeds(int(seeds), 0, self.B)
        if isinstance(seeds, torch.Tensor):
            s = seeds.to(self.device).contiguous()
        else:
            s = torch.from_numpy(np.ascontiguousarray(seeds, dtype=np.uint64).view(np.int64)).to(self.device)
        if s.numel() != self.B:
            raise DcmError("seeds must have one entry per env")
        with torch.cuda.device(self.device):
            check(self._lib.dcm_reset(self._h, _ptr(s), self._stream()))
        self._seeds = s
        return self.observe() if observe else None

    # ------------------------------------------------------------------ observe / step
    def _obs(self):
        return self._obs_static

    def observe(self, leader=None):
        li = None if leader is None else self._dev(leader, torch.int32)
        with torch.cuda.device(self.device):
            check(self._lib.dcm_observe(self._h, _ptr(self._agents), _ptr(self._tasks), _ptr(self._mask),
                                        _ptr(self._leader), _ptr(self._active), _ptr(li), self._stream()))
        return self._obs()

    def step(self, actions, leader=None, n_followers=None, followers=None, observe=True):
        """actions int32[B] (0 = depot, k = task k-1).  Optional injected choices for parity replays:
        leader int32[B] (-1 = draw), n_followers int32[B] (-1 = draw), followers int16[B,4]."""
        if (leader is None and n_followers is None and followers is None and observe and type(actions) is torch.Tensor
                and actions.dtype is torch.int32 and actions.is_cuda and actions.is_contiguous()
                and actions.device.index == self._dev_index and actions.numel() == self.B and torch.cuda.current_device() == self._dev_index):
            # the plain call of a policy in the loop (worker.py:73-76): no conversions, no device switch, no per-call allocation
            rc = self._dcm_step(self._h, actions.data_ptr(), None, None, None, *self._out_ptrs, _RAW_STREAM(self._dev_index))
            if rc != 0:
                check(rc)
            return self._obs_static
        a = self._dev(actions, torch.int32)
        li = None if leader is None else self._dev(leader, torch.int32)
        nf = None if n_followers is None else self._dev(n_followers, torch.int32)
        fo = None if followers is None else self._dev(followers, torch.int16)
        if fo is not None and tuple(fo.shape) != (self.B, _lib.FOLLOWER_COLS):
            raise DcmError("followers must be int16[B,4]")
        o = (self._agents, self._tasks, self._mask, self._leader, self._active) if observe else (None,) * 5
        with torch.cuda.device(self.device):
            check(self._lib.dcm_step(self._h, _ptr(a), _ptr(li), _ptr(nf), _ptr(fo), *[_ptr(x) for x in o],
                                     self._stream()))
        return self._obs() if observe else None

    def rollout_random(self, episodes=1, write_obs=True, max_decisions=-1):
        """Config-2 hot path: `episodes` full random-policy episodes per env in one persistent launch.
        max_decisions: int (all envs) or int64[B] (numpy/torch) -- decision budget of this call (< 0 = unlimited); an env that
        runs out of budget stays at its pending decision, the obs buffers (obs()) hold what its last decision TAKEN saw.
        Returns steps int64[B] (device tensor)."""
        steps = torch.empty((self.B,), dtype=torch.int64, device=self.device)
        o = (self._agents, self._tasks, self._mask) if write_obs else (None, None, None)
        per_env = None
        if not isinstance(max_decisions, (int, np.integer)):
            per_env = self._dev(max_decisions, torch.int64)
            if per_env.numel() != self.B:
                raise DcmError("max_decisions must be an int or int64[B]")
            max_decisions = -1
        with torch.cuda.device(self.device):
            check(self._lib.dcm_rollout_random(self._h, int(episodes), int(max_decisions), _ptr(per_env),
                                               *[_ptr(x) for x in o], _ptr(steps), self._stream()))
        return steps

    def obs(self):
        """The env's static observation buffers (written by the last observe / step / rollout_random call)."""
        return self._obs()

    # ------------------------------------------------------------------ results
    def summary(self):
        """f64[B,8]: reward, n_finished, success_rate, makespan, time_cost, waiting_time, travel_dist, efficiency."""
        out = torch.empty((self.B, 8), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            check(self._lib.dcm_summary(self._h, _ptr(out), self._stream()))
        return out

    def status(self):
        flags = torch.empty((self.B,), dtype=torch.int32, device=self.device)
        dec = torch.empty((self.B,), dtype=torch.int64, device=self.device)
        now = torch.empty((self.B,), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            check(self._lib.dcm_env_status(self._h, _ptr(flags), _ptr(dec), _ptr(now), self._stream()))
        return dict(flags=flags, decisions=dec, now=now)

    def episodes(self):
        """int32[B]: episodes finished by each env since reset()."""
        out = torch.empty((self.B,), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            check(self._lib.dcm_env_episodes(self._h, _ptr(out), self._stream()))
        return out

    def tasks_state(self):
        B, T, dev = self.B, self.T, self.device
        u8 = lambda: torch.empty((B, T), dtype=torch.uint8, device=dev)
        f8 = lambda: torch.empty((B, T), dtype=torch.float64, device=dev)
        i4 = lambda: torch.empty((B, T), dtype=torch.int32, device=dev)
        o = dict(finished=u8(), feasible=u8(), time_start=f8(), time_finish=f8(), sum_waiting_time=f8(), status=i4(),
                 n_members=i4(), n_abandoned=i4())
        with torch.cuda.device(dev):
            check(self._lib.dcm_get_tasks(self._h, *[_ptr(v) for v in o.values()], self._stream()))
        return o

    def agents_state(self):
        B, A, dev = self.B, self.A, self.device
        f8 = lambda: torch.empty((B, A), dtype=torch.float64, device=dev)
        u8 = lambda: torch.empty((B, A), dtype=torch.uint8, device=dev)
        o = dict(sum_waiting_time=f8(), travel_dist=f8(), next_decision=f8(), arrival=f8(), x=f8(), y=f8(),
                 returned=u8(), assigned=u8(), current=torch.empty((B, A), dtype=torch.int32, device=dev),
                 pending_group=torch.empty((B, A), dtype=torch.int32, device=dev))
        with torch.cuda.device(dev):
            check(self._lib.dcm_get_agents(self._h, *[_ptr(v) for v in o.values()], self._stream()))
        return o

    def task_members(self):
        """int16[B,T,member_cap]: task['members'] of every task in list order, -1 padded (env/task_env.py:80)."""
        out = torch.empty((self.B, self.T, self.member_cap), dtype=torch.int16, device=self.device)
        with torch.cuda.device(self.device):
            check(self._lib.dcm_get_members(self._h, _ptr(out), self._stream()))
        return out

    def abandoned_counts(self):
        """int16[B,A,T]: how often task t has moved agent a to its abandoned_agent list this episode (env/task_env.py:89)."""
        out = torch.empty((self.B, self.A, self.T), dtype=torch.int16, device=self.device)
        with torch.cuda.device(self.device):
            check(self._lib.dcm_get_abandoned(self._h, _ptr(out), self._stream()))
        return out

    def enable_return_log(self, cap):
        """Keep every episode's return: f64[B, cap] ring, the k-th episode an env finishes since reset() lands in column
        k mod cap (dcm_set_return_log).  summary() keeps only the last episode; with cap = episodes per rollout_random call
        all of a call's episode returns are here afterwards.  cap = 0 disables."""
        self._retlog = torch.full((self.B, int(cap)), float("nan"), dtype=torch.float64, device=self.device) if cap else None
        check(self._lib.dcm_set_return_log(self._h, _ptr(self._retlog), int(cap)))
        self.graph_epoch += 1
        return self._retlog

    # ------------------------------------------------------------------ route history (agent['route'], agent['arrival_time'])
    def enable_route_log(self, cap=64):
        """Record every agent_step of the lockstep API (reset / step): route_task[B,A,cap] (-1 = depot), route_arrival, route_len."""
        B, A, dev = self.B, self.A, self.device
        self._route = (torch.full((B, A, cap), -2, dtype=torch.int16, device=dev),
                       torch.zeros((B, A, cap), dtype=torch.float64, device=dev),
                       torch.zeros((B, A), dtype=torch.int32, device=dev))
        check(self._lib.dcm_set_route_log(self._h, *[_ptr(x) for x in self._route], int(cap)))
        self.graph_epoch += 1
        return self

    def routes(self):
        """(task[B,A,cap], arrival[B,A,cap], length[B,A]) recorded since the last reset."""
        return self._route

    # ------------------------------------------------------------------ route replay (env/task_env.py:562-599)
    def load_routes(self, routes, member_cap=8):
        """routes[b][a] = list of actions (0 = depot, k = task k-1) or None (pre_set_route stays None)."""
        B, A = self.B, self.A
        cap = max([len(r) for env in routes for r in env if r is not None] + [1])
        arr = np.zeros((B, A, cap), np.int32)
        ln = np.full((B, A), -1, np.int32)
        for b in range(B):
            for a in range(A):
                r = routes[b][a] if a < len(routes[b]) else None
                if r is not None:
                    ln[b, a] = len(r)
                    arr[b, a, :len(r)] = r
        return self.load_route_arrays(arr, ln, member_cap)

    def load_route_arrays(self, routes, route_len, member_cap=8):
        """routes int32[B,A,cap] (actions, 0-padded), route_len int32[B,A] (-1 = pre_set_route stays None): dcm_load_routes."""
        routes, route_len = np.ascontiguousarray(routes, np.int32), np.ascontiguousarray(route_len, np.int32)
        if routes.ndim != 3 or routes.shape[:2] != (self.B, self.A) or route_len.shape != (self.B, self.A):
            raise DcmError(f"routes must be int32[{self.B},{self.A},cap] and route_len int32[{self.B},{self.A}]")
        d_arr, d_ln = self._dev(routes, torch.int32), self._dev(route_len, torch.int32)
        with torch.cuda.device(self.device):
            check(self._lib.dcm_load_routes(self._h, _ptr(d_arr), _ptr(d_ln), int(routes.shape[2]), int(member_cap), self._stream()))
            torch.cuda.current_stream(self.device).synchronize()
        return self

    def set_visibility(self, initial=20, batch=20, period=10, cap=100):
        """Dynamic-arrival schedule of the reactive replay: visible = clip(now // period * batch + initial, initial, cap)
        (env/task_env.py:567; re-arm (next - 1) // batch * period, :221).  Defaults = the reference's hard-coded constants."""
        check(self._lib.dcm_set_visibility(self._h, int(initial), int(batch), int(period), int(cap)))
        return self

    def set_replay_placement(self, placement="auto"):
        """Which replay kernel execute_routes runs: "auto" = the register-resident one whenever the shape allows it (<= 128 agents,
        <= 128 live tasks, member_cap <= 8), else the general one with its replay scratch in LDS for batches of at most four envs
        per CU and in HBM otherwise; "lds" / "hbm" = always the general kernel with its scratch there (dcm_set_replay_placement).
        Results do not depend on it."""
        check(self._lib.dcm_set_replay_placement(self._h, {"auto": 0, "lds": 1, "hbm": 2}[placement]))
        return self

    def execute_routes(self, reactive=False, fields=None):
        """execute_by_route + get_episode_reward for every env; returns a dict of device tensors.
        fields: which of the optional per-task / per-agent arrays to produce (default: all; () = steps, flags, summary only)."""
        B, A, T, dev = self.B, self.A, self.T, self.device
        spec = dict(finished=((B, T), torch.uint8), time_start=((B, T), torch.float64), time_finish=((B, T), torch.float64),
                    task_wait=((B, T), torch.float64), n_members=((B, T), torch.int32), agent_wait=((B, A), torch.float64),
                    travel_dist=((B, A), torch.float64), returned=((B, A), torch.uint8))
        o = dict(steps=torch.empty((B,), dtype=torch.int64, device=dev), flags=torch.empty((B,), dtype=torch.int32, device=dev))
        for k, (shape, dt) in spec.items():
            o[k] = torch.empty(shape, dtype=dt, device=dev) if fields is None or k in fields else None
        with torch.cuda.device(dev):
            check(self._lib.dcm_execute_routes(self._h, int(bool(reactive)), *[_ptr(v) for v in o.values()], self._stream()))
        o = {k: v for k, v in o.items() if v is not None}
        o["summary"] = self.summary()
        return o

    # ------------------------------------------------------------------ snapshot (copy.deepcopy(env), worker.py:33)
    def clone_state(self):
        n = C.c_size_t()
        check(self._lib.dcm_state_bytes(self._h, C.byref(n)))
        buf = torch.empty((n.value,), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            check(self._lib.dcm_clone_state(self._h, _ptr(buf), self._stream()))
        return buf

    def restore_state(self, buf):
        with torch.cuda.device(self.device):
            check(self._lib.dcm_restore_state(self._h, _ptr(buf), self._stream()))

    def record_bytes(self):
        n = C.c_size_t()
        check(self._lib.dcm_record_bytes(self._h, C.byref(n)))
        return n.value


def device_distance(a_xy, b_xy, device="cuda:0"):
    """calculate_eulidean_distance + travel time on the device (known-answer test helper)."""
    lib = _lib.load()
    dev = torch.device(device)
    a = torch.as_tensor(np.ascontiguousarray(a_xy), dtype=torch.float64).to(dev)
    b = torch.as_tensor(np.ascontiguousarray(b_xy), dtype=torch.float64).to(dev)
    ax, ay, bx, by = a[:, 0].contiguous(), a[:, 1].contiguous(), b[:, 0].contiguous(), b[:, 1].contiguous()
    d = torch.empty_like(ax)
    t = torch.empty_like(ax)
    with torch.cuda.device(dev):
        check(lib.dcm_distance(_ptr(ax), _ptr(ay), _ptr(bx), _ptr(by), _ptr(d), _ptr(t), ax.numel(),
                               C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return d.cpu().numpy(), t.cpu().numpy()
