"""TaskEnv -- the reference's per-env method surface (env/task_env.py:8-623) on top of the HIP env.

A single-env (B=1) view with the attribute and method names worker.py / RL_test.py call, so a loop written
against the reference class (worker.py:45-87) runs unchanged:

    decision_agents, t = env.next_decision();  groups = env.get_unique_group(decision_agents)
    env.current_time = t;  env.task_update();  env.agent_update()
    ... mask = env.get_unfinished_task_mask(); env.get_current_agent_status(agent); env.get_current_task_status(agent)
    group, r = env.step(group, leader_id, action, idx);  env.task_update();  env.agent_update()
    env.finished = env.check_finished()
    reward, finished_tasks = env.get_episode_reward(MAX_TIME)

The HIP kernels fuse step + task_update + agent_update + check_finished + next event, so this adapter
answers the fine-grained calls from the already-advanced device state: task_update / agent_update are
no-ops (the device applied them), next_decision / get_unique_group report the deciding groups the device
computed, and current_time only moves forward when the caller commits it (`env.current_time = t`,
worker.py:49), which keeps the `current_time < MAX_TIME` loop test of worker.py:45 on the previous event's time
exactly like the reference (quirk Q7).  It is a compatibility surface for the reference's call pattern, not a
general re-implementation of every call order: a caller that commits a time other than the event time next_decision() returned
gets a ValueError instead of stale answers.  Use BatchedTaskEnv for throughput.

Differences a caller can observe: observations are the fp32 values the policy receives (the reference returns
fp64 and casts at worker.py:62,64); followers are drawn by the keyed choice protocol instead of numpy's RNG.
"""
import numpy as np
import torch

from . import _lib
from .batched_env import BatchedTaskEnv
from .instances import generate_instance


class TaskEnv:
    _ROUTE_CAP = 256     # route entries kept per agent (the reference's lists are unbounded; max seen at its constants: 14)

    def __init__(self, agents_range=(10, 10), tasks_range=(10, 10), traits_dim=1, max_coalition_size=3, max_duration=5,
                 seed=None, plot_figure=False, device="cuda:0", choice_seed=0, individual_selection=False):
        if traits_dim != 1:
            raise NotImplementedError("traits_dim != 1 does not work in the reference either (SURVEY.md §5)")
        if max_coalition_size > _lib.MAX_MEMBERS_WIDE:
            raise ValueError(f"max_coalition_size <= {_lib.MAX_MEMBERS_WIDE}")
        from .instances import generate_instance_ranges
        A, inst = generate_instance_ranges(agents_range, tasks_range, seed, max_coalition_size, max_duration)   # :58-71
        depot, task_xy, req, dur = inst["depot"], inst["task_xy"], inst["req"], inst["dur"]
        self._init_from_arrays(A, depot, task_xy, req, dur, device, choice_seed, individual_selection=individual_selection)

    @classmethod
    def from_arrays(cls, n_agents, depot, task_xy, req, dur, device="cuda:0", choice_seed=0, max_waiting_time=10.0,
                    individual_selection=False):
        """individual_selection=True: the run_test_IS call pattern (worker.py:159-198) -- deciders are not grouped and act
        one by one through agent_step()."""
        self = cls.__new__(cls)
        self._init_from_arrays(int(n_agents), np.asarray(depot, np.float64), np.asarray(task_xy, np.float64),
                               np.asarray(req, np.int32), np.asarray(dur, np.float64), device, choice_seed,
                               max_waiting_time, individual_selection)
        return self

    def _init_from_arrays(self, A, depot, task_xy, req, dur, device, choice_seed, max_waiting_time=10.0,
                          individual_selection=False):
        self.agents_num, self.tasks_num = A, len(req)
        self._individual_selection = bool(individual_selection)
        self.max_waiting_time = float(max_waiting_time)
        self.reactive_planning = False
        self.dt = 0.1
        self._inst = (depot.copy(), task_xy.copy(), req.copy(), dur.copy())
        self._env = BatchedTaskEnv(1, A, self.tasks_num, device=device, max_waiting_time=self.max_waiting_time,
                                   individual_selection=individual_selection,
                                   member_cap=_lib.MAX_MEMBERS_WIDE if int(np.max(req)) > _lib.MAX_MEMBERS else _lib.MAX_MEMBERS)
        self._env.load_instances(depot[None], task_xy[None], req[None], dur[None])
        self._env.enable_route_log(cap=self._ROUTE_CAP)         # agent['route'] / agent['arrival_time'] (:95-96,314,318)
        self._seed = np.array([choice_seed], dtype=np.uint64)
        self._preset = [None] * A                      # agent['pre_set_route'], :595-599
        self._replay_summary = None
        self.depot = {"location": depot.copy(), "members": [], "ID": -1}
        self.clear_decisions()

    # ------------------------------------------------------------------ reset (env/task_env.py:116-140)
    def reset(self, test_env=None, seed=None):
        """:116-127.  test_env = (task_dic, agent_dic, depot) in the reference's dict layout (what RL_test.py:36-42 and
        baselines/CTAS-D.py:60-66 pass after unpickling a test-set env): the instance is read out of the dicts
        (location / requirements / time per task, the agent count, the depot location) and loaded onto the device."""
        if seed is not None:
            self._seed = np.array([seed], dtype=np.uint64)
        if test_env is not None:
            from .instances import instance_from_dicts
            A, inst = instance_from_dicts(*test_env)
            dep, xy, req, dur = inst["depot"], inst["task_xy"], inst["req"], inst["dur"]
            old = self._env
            self._init_from_arrays(A, dep, xy, req, dur, str(old.device), int(self._seed[0]), self.max_waiting_time,
                                   self._individual_selection)
            old.close()
            return
        self.clear_decisions()

    def clear_decisions(self):
        self._preset = [None] * self.agents_num
        self._replay_summary = None
        self._env.reset(self._seed, observe=False)
        self.finished = False
        self._visible_time = 0.0
        self._dirty = True

    # ------------------------------------------------------------------ device state cache
    def _sync(self):
        if not self._dirty:
            return
        st = self._env.status()
        self._flags = int(st["flags"][0])
        self._now = float(st["now"][0])
        self._decisions = int(st["decisions"][0])
        ag = {k: v[0].cpu().numpy() for k, v in self._env.agents_state().items()}
        tk = {k: v[0].cpu().numpy() for k, v in self._env.tasks_state().items()}
        self._ag, self._tk = ag, tk
        obs = self._env.observe()
        self._leader_dev = int(obs.leader[0])
        rt, ra, rl = (x[0].cpu().numpy() for x in self._env.routes())
        if (rl > self._ROUTE_CAP).any():
            raise RuntimeError(f"an agent's route has more than {self._ROUTE_CAP} entries (TaskEnv._ROUTE_CAP)")
        self._routes = [(rt[a, :rl[a]].astype(np.int64).tolist(), ra[a, :rl[a]].tolist()) for a in range(self.agents_num)]
        mem = self._env.task_members()[0].cpu().numpy()
        self._members = [[int(m) for m in mem[t] if m >= 0] for t in range(self.tasks_num)]
        ab = self._env.abandoned_counts()[0].cpu().numpy()
        self._abandoned = [[a for a in range(self.agents_num) for _ in range(int(ab[a, t]))] for t in range(self.tasks_num)]
        if self._flags & (_lib.FLAG_BAD_ACTION | _lib.FLAG_OVERFLOW | _lib.FLAG_BAD_LEADER | _lib.FLAG_BAD_INSTANCE):
            raise RuntimeError(f"env error flags {self._flags:#x}")
        if self._flags & _lib.FLAG_TRUNCATED:
            raise RuntimeError("every agent is at the depot while a task can never become feasible: the reference "
                               "loops forever here (SURVEY.md §5); the device env truncated the episode")
        self._dirty = False

    @property
    def _done(self):
        self._sync()
        return bool(self._flags & _lib.FLAG_DONE)

    # current_time: the caller commits the new event time (worker.py:49)
    @property
    def current_time(self):
        self._sync()
        return self._now if self._done else self._visible_time

    @current_time.setter
    def current_time(self, t):
        # The reference recomputes task_update / agent_update from whatever time the caller has set (env/task_env.py:245-281);
        # here they were applied on the device at the event time next_decision() reports.  Committing any OTHER time would make
        # the following (no-op) task_update / agent_update silently stale: refuse it instead.
        self._sync()
        if not self._done and float(t) != self._now and float(t) != self._visible_time:
            raise ValueError(f"TaskEnv.current_time = {float(t)!r}: only the event time returned by next_decision() "
                             f"({self._now!r}) can be committed; the device env advances time itself (use BatchedTaskEnv / the "
                             f"reference's call order worker.py:45-87)")
        self._visible_time = float(t)

    @property
    def agent_dic(self):
        self._sync()
        a = self._ag
        # route / arrival_time: the agent's visiting history (task ids, -1 = depot) and the arrival time of every entry
        # (env/task_env.py:95-96, appended by agent_step :314,:318; read by worker.py:244-251 and generate_traj :375-418)
        return {i: {"ID": i, "location": np.array([a["x"][i], a["y"][i]]), "returned": bool(a["returned"][i]),
                    "assigned": bool(a["assigned"][i]), "next_decision": float(a["next_decision"][i]),
                    "travel_dist": float(a["travel_dist"][i]), "sum_waiting_time": float(a["sum_waiting_time"][i]),
                    "route": list(self._routes[i][0]), "arrival_time": list(self._routes[i][1]),
                    "pre_set_route": self._preset[i], "velocity": 0.2, "depot": self.depot["location"]}
                for i in range(self.agents_num)}

    @property
    def task_dic(self):
        self._sync()
        t = self._tk
        _, xy, req, dur = self._inst
        return {i: {"ID": i, "requirements": np.array([req[i]]), "location": xy[i].copy(), "time": float(dur[i]),
                    "feasible_assignment": bool(t["feasible"][i]), "finished": bool(t["finished"][i]),
                    "time_start": float(t["time_start"][i]), "time_finish": float(t["time_finish"][i]),
                    "status": np.array([int(t["status"][i])]), "sum_waiting_time": float(t["sum_waiting_time"][i]),
                    # members: ordered list of agent ids (:78); abandoned_agent: every agent the task gave up on, once per
                    # abandonment (:89; grouped by ascending agent id here -- the reference appends in event order, and
                    # only membership / multiplicity are ever read, :357,:363)
                    "members": list(self._members[i]), "abandoned_agent": list(self._abandoned[i]),
                    "n_members": int(t["n_members"][i]), "n_abandoned": int(t["n_abandoned"][i])}
                for i in range(self.tasks_num)}

    @staticmethod
    def get_matrix(dictionary, key):  # env/task_env.py:150-159
        return [v[key] for v in dictionary.values()]

    # ------------------------------------------------------------------ event queries (env/task_env.py:283-298)
    def _pending_groups(self):
        """Groups of the current event that still have to decide, in the device's (x, y) order."""
        self._sync()
        if self._done:
            return []
        pg = self._ag["pending_group"]  # 0 = not deciding, g = index of its group (np.unique order, :293)
        return [[a for a in range(self.agents_num) if pg[a] == g] for g in sorted(set(pg[pg > 0].tolist()))]

    def next_decision(self):
        self._sync()
        if self._done:
            arr = [self._ag["arrival"][a] if self._ag["current"][a] != -2 else 0 for a in range(self.agents_num)]
            return [], max(arr)
        ids = np.array(sorted(a for g in self._pending_groups() for a in g), dtype=np.int64)
        return ids, self._now

    def get_unique_group(self, agents):
        want = set(int(a) for a in agents)
        return [[a for a in g if a in want] for g in self._pending_groups() if any(a in want for a in g)]

    def task_update(self):
        """Applied on the device inside dcm_reset / dcm_step at the reference's call sites (worker.py:50,74): a no-op here,
        valid because the only times a caller can commit are the device's own event times (current_time setter)."""
        return []

    def agent_update(self):  # idem (worker.py:51,76)
        return None

    # ------------------------------------------------------------------ observations (env/task_env.py:165-200)
    def _observe(self, leader):
        return self._env.observe(leader=np.array([leader], np.int32))

    def get_unfinished_task_mask(self):
        self._sync()
        t = self._tk
        return np.logical_not((t["feasible"] == 0) & (t["status"] > 0))

    def get_current_agent_status(self, agent):
        return self._observe(agent["ID"]).agents[0].cpu().numpy().astype(np.float64)

    def get_current_task_status(self, agent):
        return self._observe(agent["ID"]).tasks[0].cpu().numpy().astype(np.float64)

    # ------------------------------------------------------------------ step (env/task_env.py:326-342)
    def step(self, group, leader_id, action, current_action_index=0):
        self._sync()
        before = self._ag
        pos0 = np.stack([before["x"], before["y"]], 1).copy()
        gid = int(before["pending_group"][leader_id])
        n_before = self._decisions
        self._env.step(np.array([action], np.int32), leader=np.array([leader_id], np.int32), observe=False)
        self._dirty = True
        self._sync()
        # members of this step = agents of the group that left the pending set; once the event is over
        # (new event or terminal) the whole remaining group has left
        same_event = (not self._done) and self._now == self._visible_time and \
            any(int(self._ag["pending_group"][a]) == gid for a in group if a != leader_id)
        moved = [a for a in group if a == leader_id or not same_event or int(self._ag["pending_group"][a]) != gid]
        if self._decisions == n_before:
            raise RuntimeError("leader is not in the deciding group")
        target = self.depot["location"] if action == 0 else self._inst[1][action - 1]
        tt = [float(np.linalg.norm(pos0[a] - target)) / 0.2 for a in moved] or [0.0]
        for a in moved:
            if a in group:
                group.remove(a)
        return group, -float(np.mean(tt))

    def agent_step(self, agent_id, task_id):
        """Individual selection (worker.py:186): the agent acts alone, no followers."""
        self._sync()
        self._env.step(np.array([task_id], np.int32), leader=np.array([agent_id], np.int32),
                       n_followers=np.array([0], np.int32), followers=np.full((1, 4), -1, np.int16), observe=False)
        self._dirty = True

    # ------------------------------------------------------------------ termination (env/task_env.py:366-373,420-425)
    def check_finished(self):
        self._sync()
        return bool(self._flags & _lib.FLAG_FINISHED)

    def get_episode_reward(self, max_time=100):
        self._sync()
        if not self._done:
            raise RuntimeError("episode still running: the device env computes the reward at its terminal state")
        return -self._now, [bool(f) for f in self._tk["finished"]]

    def perf_metrics(self):
        """worker.py:103-108."""
        sm = self._replay_summary if self._replay_summary is not None else self._env.summary()[0].cpu().numpy()
        return dict(success_rate=sm[2], makespan=sm[3], time_cost=sm[4], waiting_time=sm[5], travel_dist=sm[6],
                    efficiency=sm[7])

    # ------------------------------------------------------------------ route replay (env/task_env.py:562-599)
    def pre_set_route(self, routes, agent_id):
        """:595-599: set, or extend, the preset action list of one agent (0 = depot, k = task k-1)."""
        cur = self._preset[agent_id]
        self._preset[agent_id] = list(routes) if not cur else cur + list(routes)

    def execute_by_route(self, path="./", method=0, plot_figure=False):
        """:562-593 on the device (dcm_load_routes + dcm_execute_routes; max_waiting_time 100, cut-off 200, dynamic task
        visibility when `reactive_planning` is set).  Afterwards current_time, get_episode_reward, task_dic / agent_dic
        (finished, time_start, time_finish, sum_waiting_time, travel_dist, returned) describe the replayed episode, as
        baselines/CTAS-D.py:83-94 reads them.  Raises TypeError where the reference does (:220, pre_set_route None)."""
        self._env.load_routes([self._preset], member_cap=min(32, max(8, self.agents_num)))
        out = self._env.execute_routes(reactive=bool(self.reactive_planning))
        flags = int(out["flags"][0])
        if flags & _lib.FLAG_TYPE_ERROR:
            raise TypeError("'NoneType' object is not subscriptable")       # the reference's own failure, :220
        if flags & (_lib.FLAG_OVERFLOW | _lib.FLAG_BAD_ACTION):
            raise RuntimeError(f"route replay error flags {flags:#x} (more visitors on one task than member_cap, or a bad action id)")
        if flags & _lib.FLAG_TRUNCATED:
            raise RuntimeError("the reference never terminates on these routes (every agent idle while tasks stay open)")
        g = lambda k: out[k][0].cpu().numpy()
        sm = out["summary"][0].cpu().numpy()
        A, T = self.agents_num, self.tasks_num
        nanA, zerA = np.full(A, np.nan), np.zeros(A)
        self._ag = dict(x=nanA, y=nanA, returned=g("returned"), assigned=zerA, next_decision=nanA, arrival=nanA,
                        travel_dist=g("travel_dist"), sum_waiting_time=g("agent_wait"), current=np.full(A, -1), pending_group=zerA)
        fin = g("finished")
        self._tk = dict(feasible=fin.copy(), finished=fin, time_start=g("time_start"), time_finish=g("time_finish"),
                        status=np.zeros(T, np.int32), sum_waiting_time=g("task_wait"), n_members=g("n_members"),
                        n_abandoned=np.zeros(T, np.int32))
        # (the replay kernel keeps no visiting history / member lists: these views are empty after execute_by_route)
        self._routes = [([], []) for _ in range(A)]
        self._members = [[] for _ in range(T)]
        self._abandoned = [[] for _ in range(T)]
        self._now = float(sm[3])
        self._flags = flags | _lib.FLAG_DONE
        self._replay_summary = sm
        self._dirty = False
        self.finished = bool(flags & _lib.FLAG_FINISHED)
        self.max_waiting_time = 100.0                                        # :564
        return self._now
