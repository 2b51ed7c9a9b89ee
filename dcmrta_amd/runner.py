"""BatchedRunner -- the rollout side of runner.py / worker.py for a whole batch of envs on one GPU.

What one Ray actor of the reference does for ONE env per `job` (runner.py:58-71 -> worker.py:41-112) this class does
for B envs at once: a sampled episode per env recorded as the 9-slot experience the learner consumes
(driver.py:135-164), the greedy self-critic twin rollout on the same instances (worker.py:89-92,200-235), the
advantage (worker.py:91-101, GAMMA=1 so every decision of an episode carries the episode advantage) and the six
perf metrics (worker.py:103-108).  The env work is the HIP path (BatchedTaskEnv); the policy is whatever module the
caller supplies (`net_factory`, e.g. the reference's own AttentionNet) -- by default the stand-in of dcmrta_amd.policy.

Every rollout is one captured HIP graph per decision (policy forward -> action -> experience store -> dcm_step with the
fused observation of the next decision; dcmrta_amd/graph_rollout.py) replayed without a host sync per decision: the host
looks at the `active` flags once every `check_every` batched steps.

job() keeps the reference signature and return shape:
    jobResults : list of 9 sequences; torch.stack(jobResults[k]) gives (N,A,6), (N,T+1,5), (N,1) int64, (N,T+1) bool,
                 (N,1), (N,1,1) int64, (N,1) -- N = decisions of all B episodes, episode-major order
    metrics    : dict with success_rate, makespan, time_cost, waiting_time, travel_dist, efficiency (means over the batch)
    info       : {"id", "episode_number"}
"""
import collections

import numpy as np
import torch

from . import _lib
from .batched_env import BatchedTaskEnv
from .choice import env_seeds
from .graph_rollout import GraphedRollout
from .instances import generate_batch, generate_batch_ranges

METRIC_KEYS = ("success_rate", "makespan", "time_cost", "waiting_time", "travel_dist", "efficiency")
_PRECISIONS = {"fp32": None, "bf16": torch.bfloat16, "fp16": torch.float16}


class EnvError(RuntimeError):
    """An env of the batch was frozen by the device (bad action / member overflow / bad leader): its results are undefined."""


class BatchedRunner:
    def __init__(self, metaAgentID=0, n_envs=256, device="cuda:0", net_factory=None, base_seed=0, max_steps=None, gamma=1.0,
                 rollout_precision="fp32", check_every=8, use_graph=True, cache_shapes=8, buckets="auto", episode_stride=None,
                 env_offset=0, strict_mask=None, twin_rollout="auto", tune_gemms=False, gemm_tuning_file=None):
        """rollout_precision "bf16" / "fp16": the rollouts (sampled, greedy twin, evaluation) run a low-precision shadow
        of localNetwork (net.rollout_copy(dtype), refreshed after every weight update); needs a net that offers
        rollout_copy / sync_rollout_copy (the stand-in does).  max_steps: capacity of the experience record in batched
        steps (default 6 (A + T) + 64, ~3x the longest episode seen at the reference's constants).  buckets: e.g.
        (1.0, 0.5, 0.25) -- the policy runs only on the envs still active once half / three quarters of the episodes are
        over (GraphedRollout compaction; worthwhile when the forward is much more expensive than the env step).  "auto" (default):
        (1.0, 0.5, 0.25, 0.125) from 1024 envs on -- a 4096-env job of the attention policy takes 1.8 instead of 2.2 s -- none
        below (the loop is launch-bound there and three more graphs per shape and rollout kind only cost capture time).
        tune_gemms: before the rollout graphs of a new (net, shape) are captured, the policy is run once at every batch size the
        graphs will use with PyTorch's TunableOp tuning switched on (stock torch.cuda.tunable: picks the fastest rocBLAS /
        hipBLASLt solution per GEMM shape; 8.0 -> 6.8 ms per fp32 forward at 4096 x 20A/50T) and the selection is kept for the
        graphs.  Tuning a shape takes about a minute and a half the first time; the results accumulate in `gemm_tuning_file`
        (default: PyTorch's own tunableop_results file) across rounds and runs, so it pays for long trainings, not for one job."""
        self.metaAgentID = metaAgentID
        self.device = torch.device(device)
        self.B = int(n_envs)
        # job(episodeNumber) plays the instances / seeds [episodeNumber * episode_stride + env_offset, ... + n_envs): with the
        # default stride n_envs the blocks of successive jobs are disjoint; runners that share one env budget of unequal shards
        # (dist.shard_range) use a common stride >= the largest shard (ray_compat) or the whole budget plus their shard offset
        # (dist_runner), so that no two of them ever play the same instance
        # strict_mask: freeze (and report, EnvError) an env whose policy picks a masked task instead of simulating the action
        # the way the reference's TaskEnv.step does.  None (default): strict for the sampled / greedy rollouts of job() and
        # testing() -- a masked pick there can only come from a broken policy (NaN logits of a low-precision shadow make argmax
        # return the NaN index) and must not be fed to REINFORCE silently -- and the reference's semantics (any action is
        # simulated) for run_test, whose argmax(logp.exp() * ~mask) can legitimately land on a masked index (worker.py:140)
        self.strict_mask = None if strict_mask is None else bool(strict_mask)
        # twin_rollout: job() plays the sampled episode and its greedy self-critic twin (worker.py:89) of every env in ONE batch
        # of 2 n_envs (rows [0, B) sample, rows [B, 2B) take the argmax of the same net on the same instance and seed): one
        # policy forward, one graph replay and one tail per decision instead of two -- what pays when the loop is launch-bound
        # (small batches) or the tail of finishing episodes is long
        # ("auto": below 1024 envs -- 8.5e4 -> 1.4e5 recorded decisions/s at 256 envs; from there on the forward is throughput-bound
        #  and two batches of B cost the same as one of 2 B)
        self.twin_rollout = (self.B < 1024) if twin_rollout == "auto" else bool(twin_rollout)
        self.tune_gemms = bool(tune_gemms)
        if self.tune_gemms:
            import torch.cuda.tunable as tunable
            tunable.enable(True)
            tunable.tuning_enable(False)
            if gemm_tuning_file:
                tunable.set_filename(str(gemm_tuning_file))
        self.episode_stride = int(episode_stride) if episode_stride is not None else self.B
        self.env_offset = int(env_offset)
        if self.episode_stride < self.B:
            raise ValueError("episode_stride must be >= n_envs")
        self.base_seed = int(base_seed)
        self.max_steps = None if max_steps is None else int(max_steps)
        self.gamma = float(gamma)            # GAMMA, parameters.py:6 (1 in the reference)
        self.check_every, self.use_graph = int(check_every), bool(use_graph)
        if isinstance(buckets, str):
            if buckets != "auto":
                raise ValueError('buckets: "auto", None or a tuple of fractions')
            buckets = (1.0, 0.5, 0.25, 0.125) if self.B >= 1024 else None
        self.buckets = tuple(buckets) if buckets else None
        if net_factory is None:
            from .policy import AttentionNet
            net_factory = lambda: AttentionNet(6, 5, 128)  # AGENT_INPUT_DIM, TASK_INPUT_DIM, EMBEDDING_DIM (parameters.py:29-31)
        self.localNetwork = net_factory().to(self.device)     # runner.py:18-19
        self.localBaseline = net_factory().to(self.device)    # runner.py:20-21
        if rollout_precision not in _PRECISIONS:
            raise ValueError(f"rollout_precision must be one of {sorted(_PRECISIONS)}")
        self.rollout_precision = rollout_precision
        self._shadow = None
        if _PRECISIONS[rollout_precision] is not None:
            if not hasattr(self.localNetwork, "rollout_copy"):
                raise ValueError("rollout_precision != 'fp32' needs a net with rollout_copy() / sync_rollout_copy()")
            self._shadow = self.localNetwork.rollout_copy(_PRECISIONS[rollout_precision])
        self._cache = collections.OrderedDict()   # (A, T, individual_selection) -> {"env", "graphs"}
        self._cache_shapes = int(cache_shapes)
        self._env = None
        self.timing = {}
        self.last = {}

    # ------------------------------------------------------------------ weights (runner.py:23-30)
    def get_weights(self):
        return self.localNetwork.state_dict()

    def set_weights(self, weights):
        self.localNetwork.load_state_dict(weights)          # in place: captured graphs keep reading the same tensors
        if self._shadow is not None:
            self.localNetwork.sync_rollout_copy(self._shadow)

    def set_baseline_weights(self, weights):
        self.localBaseline.load_state_dict(weights)

    def _rollout_net(self):
        return self._shadow if self._shadow is not None else self.localNetwork

    # ------------------------------------------------------------------ env management
    def _slot(self, A, T, individual_selection=False, n_envs=None, strict=True):
        """Env + captured graphs of one batch shape.  Training draws a new (agents_num, tasks_num) every round
        (driver.py:114-115), so the last few shapes are kept.  strict: what strict_mask=None resolves to for this use."""
        strict = bool(strict) if self.strict_mask is None else self.strict_mask
        key = (A, T, bool(individual_selection), int(n_envs or self.B), strict)
        slot = self._cache.get(key)
        if slot is None:
            # (under the capture lock: dcm_create allocates and may set kernel attributes -- not while another actor thread of
            #  this process is in the middle of a stream capture)
            from .graph_rollout import CAPTURE_LOCK
            with CAPTURE_LOCK:
                slot = dict(env=BatchedTaskEnv(key[3], A, T, device=str(self.device), individual_selection=individual_selection,
                                               strict_mask=strict), graphs={})
            self._cache[key] = slot
            while len(self._cache) > self._cache_shapes:
                _, old = self._cache.popitem(last=False)
                old["graphs"].clear()
                old["env"].close()
        else:
            self._cache.move_to_end(key)
        self._env = slot["env"]
        return slot

    def _get_env(self, A, T):
        return self._slot(A, T)["env"]

    # ------------------------------------------------------------------ one batched episode (worker.py:45-87)
    @staticmethod
    def _select(mode):
        if mode == "sample":
            # worker.py:70: Categorical(logp.exp()).sample().  Drawn as argmax(p / q), q ~ Exp(1) (the "exponential race",
            # which is also what torch.multinomial does for one sample) written out on the log-probabilities: no exp(), graph-
            # capturable (torch.multinomial on these probabilities faults under HIP-graph replay at B = 4096 on ROCm 7.2), and
            # a masked action (logp ~ -1e4, attention.py:76) can never win: -log q <= 88 in fp32
            return lambda logp, mask: torch.argmax(logp - torch.empty_like(logp).exponential_(1.0).log(), dim=1)
        if mode == "greedy":      # worker.py:228
            return lambda logp, mask: torch.argmax(logp, dim=1)
        if mode == "twin":        # rows below `half` sample, the others take the argmax (see twin_rollout)
            raise ValueError("the twin selection needs the row split: use _twin_policy")
        if mode == "test":        # worker.py:140 / :185
            return lambda logp, mask: torch.argmax(logp.exp() * ~mask, dim=1)
        raise ValueError(mode)

    def _check_flags(self, env, what):
        """The device freezes an env whose action is out of range or -- on a strict handle: every rollout of job() / testing()
        unless strict_mask=False was asked for -- masked (BAD_ACTION), that would list more members than the handle's slots
        (OVERFLOW), or whose injected leader is not deciding (BAD_LEADER), and never computes its terminal row: refuse to
        average NaN rewards into the batch.  On a non-strict handle (run_test) a masked action is simulated like the
        reference's TaskEnv.step does (env/task_env.py:326-342).  Truncated episodes (zero-decider guard) are counted."""
        flags = env.status()["flags"]
        bad = (flags & (_lib.FLAG_BAD_ACTION | _lib.FLAG_OVERFLOW | _lib.FLAG_BAD_LEADER | _lib.FLAG_BAD_INSTANCE)) != 0
        n_bad = int(bad.sum())
        if n_bad:
            first = int(torch.nonzero(bad)[0])
            raise EnvError(f"{what}: {n_bad} of {env.B} envs were frozen by the device (first: env {first}, flags "
                           f"{int(flags[first]):#x}): the policy chose a masked / out-of-range action or a task overflowed; "
                           f"check that the net puts probability 0 on masked actions (attention.py:74-76)")
        return int(((flags & _lib.FLAG_TRUNCATED) != 0).sum())

    @torch.no_grad()
    def rollout(self, net, slot, seeds, mode, record):
        """One episode per env under `net`; mode "sample" | "greedy" | "test" | "twin" (first half of the envs samples, second
        half is greedy).  Returns (summary[B,8], rec, n_steps): rec = dict of [n_steps, B, ...] experience tensors (views of the
        graph's static buffers: consume them before the next recorded rollout of the same shape) or None."""
        env = slot["env"]
        if mode == "twin":
            # 1 for the sampling rows, 0 for the greedy ones: argmax(logp - gate * log(q)) is the exponential race on the
            # former and the plain argmax on the latter; `rows` = the env indices a compacted graph runs the policy on
            gate = slot.get("twin_gate")
            if gate is None:      # built once per slot, on the device (no host tensor + synchronous copy per job)
                gate = slot["twin_gate"] = torch.cat([torch.ones(env.B // 2, 1, device=env.device),
                                                      torch.zeros(env.B - env.B // 2, 1, device=env.device)])

            def policy(obs, rows=None):
                logp = net(obs.tasks, obs.agents, obs.mask)
                g = gate if rows is None else gate.index_select(0, rows)
                return torch.argmax(logp - g.to(logp.dtype) * torch.empty_like(logp).exponential_(1.0).log(), dim=1)
            policy.wants_rows = True
        else:
            select = self._select(mode)

            def policy(obs):
                return select(net(obs.tasks, obs.agents, obs.mask), obs.mask)
        if not self.use_graph:
            return self._rollout_eager(policy, env, seeds, record)
        key = (id(net), mode, bool(record))
        g = slot["graphs"].get(key)
        if g is None:
            g = GraphedRollout(env, policy, check_every=self.check_every, record=record, capacity=self.max_steps,
                               buckets=self.buckets)
            slot["graphs"][key] = g
            if self.tune_gemms and (id(net), "tuned") not in slot:
                self._tune(net, env, seeds, g.sizes)
                slot[(id(net), "tuned")] = True
        summary, n = g.run(seeds)
        rec = {k: v[:n] for k, v in g.rec.items()} if record else None
        return summary, rec, n

    @torch.no_grad()
    def _tune(self, net, env, seeds, sizes):
        """One forward per batch size with TunableOp tuning on -- never inside a graph capture -- then tuning off again (the
        selection stays in use)."""
        import torch.cuda.tunable as tunable
        from .graph_rollout import CAPTURE_LOCK
        with CAPTURE_LOCK:
            obs = env.reset(seeds)
            tunable.tuning_enable(True)
            try:
                for n in sizes:
                    net(obs.tasks[:n], obs.agents[:n], obs.mask[:n])
                torch.cuda.synchronize(env.device)
            finally:
                tunable.tuning_enable(False)

    def _rollout_eager(self, policy, env, seeds, record):
        """The same loop as plain launches with a host sync per decision (debugging / comparison)."""
        obs = env.reset(seeds)
        cap = self.max_steps if self.max_steps is not None else 6 * (env.A + env.T) + 64
        rec = [] if record else None
        s = 0
        while bool(obs.active.any()):
            if s >= cap:
                raise RuntimeError("episode longer than max_steps")
            action = policy(obs)
            if record:
                rec.append(dict(agents=obs.agents.clone(), tasks=obs.tasks.clone(), mask=obs.mask.clone(),
                                action=action.to(torch.int64), leader=obs.leader.to(torch.int64), active=obs.active.clone()))
            obs = env.step(action.to(torch.int32))
            s += 1
        out = {k: torch.stack([r[k] for r in rec]) for k in rec[0]} if rec else None
        return env.summary(), out, s

    def _experience(self, rec, n_steps, reward, advantage, as_lists):
        """9-slot buffer of worker.py:42,77-83,91-101 for all episodes, episode-major."""
        B = reward.shape[0]
        if rec is None or n_steps == 0:       # no env took a decision
            A, T = self._env.A, self._env.T
            dev = reward.device
            e = lambda *shape, dtype=torch.float32: torch.empty((0,) + shape, dtype=dtype, device=dev)
            slots = [e(A, 6), e(T + 1, 5), e(1, dtype=torch.int64), e(T + 1, dtype=torch.bool), e(1), e(1, 1, dtype=torch.int64),
                     e(1), [], []]
            return [list(x.unbind(0)) if (as_lists and isinstance(x, torch.Tensor)) else x for x in slots]
        act = rec["active"][:n_steps].t()                                        # [B,S]

        def pick(x):
            return x[:n_steps].transpose(0, 1)[act]                              # [N,...] env-major, step order
        agents, tasks, mask = pick(rec["agents"]), pick(rec["tasks"]), pick(rec["mask"])
        action = pick(rec["action"]).unsqueeze(1)                                # (N,1) int64, slot 2
        agent_id = pick(rec["leader"]).view(-1, 1, 1)                            # (N,1,1) int64, slot 5
        counts = act.sum(1)                                                      # decisions per episode
        env_of = torch.repeat_interleave(torch.arange(B, device=act.device), counts)
        last = torch.cumsum(counts, 0) - 1
        rew = torch.zeros((agents.shape[0], 1), dtype=torch.float32, device=agents.device)
        rew[last[counts > 0], 0] = reward[counts > 0].to(torch.float32)          # slot 4: 0 except the last decision (:81,:91)
        # slot 6: discount(x, GAMMA) of a vector that is 0 except for the episode advantage at its last decision
        # (worker.py:14-15,92-101) = GAMMA^(steps to the end) * advantage; GAMMA = 1 gives the advantage everywhere
        adv = advantage.to(torch.float32)[env_of].unsqueeze(1)
        if self.gamma != 1.0:
            to_end = (last[env_of] - torch.arange(agents.shape[0], device=agents.device)).to(torch.float32)
            adv = adv * torch.pow(torch.tensor(self.gamma, device=agents.device), to_end).unsqueeze(1)
        slots = [agents, tasks, action, mask, rew, agent_id, adv, [], []]
        if as_lists:
            slots = [list(x.unbind(0)) if isinstance(x, torch.Tensor) else x for x in slots]
        return slots

    # ------------------------------------------------------------------ runner.py:58-71
    @staticmethod
    def _is_range(x):
        return isinstance(x, (tuple, list)) and int(x[0]) != int(x[1])

    def job(self, global_weights, baseline_weights, episodeNumber, agents_num, tasks_num, as_lists=False):
        """agents_num / tasks_num: ints as driver.py:114-117 passes them, or (lo, hi) ranges as Worker's defaults
        (worker.py:26) -- then every env draws its own sizes (env/task_env.py:58-65) and the batch is ragged: the
        experience rows are padded to (hi_A, hi_T + 1) with -1 rows / True mask (worker.py:253-261)."""
        self.set_weights(global_weights)
        self.set_baseline_weights(baseline_weights)
        A = int(agents_num[1] if isinstance(agents_num, (tuple, list)) else agents_num)
        T = int(tasks_num[1] if isinstance(tasks_num, (tuple, list)) else tasks_num)
        slot = self._slot(A, T, n_envs=2 * self.B if self.twin_rollout else None)
        env = slot["env"]
        first = self.first_env(episodeNumber)
        ragged = self._is_range(agents_num) or self._is_range(tasks_num)
        if ragged:
            inst = generate_batch_ranges(range(self.base_seed + first, self.base_seed + first + self.B),
                                         tuple(agents_num) if isinstance(agents_num, (tuple, list)) else int(agents_num),
                                         tuple(tasks_num) if isinstance(tasks_num, (tuple, list)) else int(tasks_num))
        else:
            inst = generate_batch(self.B, A, T, base_seed=self.base_seed, first=first)   # worker.py:32
        net = self._rollout_net()
        self._set_padding_hint(net, ragged)
        seeds = env_seeds(self.base_seed, first, self.B)
        if self.twin_rollout:
            B = self.B
            env.load_instances(**{k: (np.concatenate([v, v]) if isinstance(v, np.ndarray) else v) for k, v in inst.items()})
            summary2, rec2, n_steps = self.rollout(net, slot, np.concatenate([seeds, seeds]), "twin", record=True)
            truncated = self._check_flags(env, "sampled + greedy twin rollout")
            summary, greedy_summary = summary2[:B], summary2[B:]
            rec = {k: v[:, :B] for k, v in rec2.items()}
            grec = {k: v[:, B:] for k, v in rec2.items()} if self.keep_greedy_record else None
            g_steps = n_steps
        else:
            env.load_instances(**inst)
            summary, rec, n_steps = self.rollout(net, slot, seeds, "sample", record=True)             # run_episode
            truncated = self._check_flags(env, "sampled rollout")
            greedy_summary, grec, g_steps = self.rollout(net, slot, seeds, "greedy", record=self.keep_greedy_record)  # baseline_test :89
            truncated += self._check_flags(env, "greedy baseline rollout")
        reward, greedy_reward = summary[:, 0], greedy_summary[:, 0]
        advantage = reward - greedy_reward                                         # worker.py:92
        jobResults = self._experience(rec, n_steps, reward, advantage, as_lists)
        m = summary[:, 2:8].mean(0).cpu().numpy()
        metrics = {k: float(m[i]) for i, k in enumerate(METRIC_KEYS)}
        info = {"id": self.metaAgentID, "episode_number": episodeNumber}
        if truncated:
            info["truncated_episodes"] = truncated
        # (rec: the sampled episodes' [n_steps, B, ...] record -- views of the graph's static buffers, valid until the next recorded
        #  rollout of this shape)
        self.last = dict(summary=summary, greedy_summary=greedy_summary, n_steps=n_steps, greedy_steps=g_steps, rec=rec,
                         greedy_rec=grec, truncated=truncated)
        return jobResults, metrics, info

    def first_env(self, episodeNumber):
        """Index of the first instance / seed job(episodeNumber) plays."""
        return int(episodeNumber) * self.episode_stride + self.env_offset

    keep_greedy_record = False    # tests set it to replay the greedy twin through the oracle

    @staticmethod
    def _set_padding_hint(net, ragged):
        if hasattr(net, "assume_no_padding"):
            net.assume_no_padding = not ragged

    # ------------------------------------------------------------------ Worker.run_test / run_test_IS (worker.py:114-198)
    @torch.no_grad()
    def run_test(self, instances, n_agents=None, individual_selection=False, seeds=None):
        """Greedy evaluation of the local network on given instances, as RL_test.py:36-51 does one env at a time.

        instances: the load_instances keyword dict (depot[N,2], task_xy[N,T,2], req[N,T], dur[N,T], optionally n_agents /
        n_tasks for a ragged set) -- e.g. instances.load_instances_npz or instances.batch_from_dicts of unpickled test-set
        envs.  n_agents: agents per env (required unless the dict carries its own per-env `n_agents`).  Action =
        argmax(logp.exp() * ~mask) (worker.py:140,185).  individual_selection=True is run_test_IS: the deciders of an event
        act one by one in ascending id without grouping.  Returns {metric: float64[N]} with the six keys of
        worker.py:146-151 plus "reward"."""
        N, T = instances["req"].shape
        if n_agents is None and "n_agents" not in instances:
            raise ValueError("run_test: give n_agents (agents per env) or a per-env `n_agents` array in `instances`")
        A = int(n_agents if n_agents is not None else np.max(instances["n_agents"]))
        ss = env_seeds(self.base_seed, 0, N) if seeds is None else np.asarray(seeds, dtype=np.uint64)
        slot = self._slot(A, T, individual_selection=individual_selection, n_envs=N, strict=False)
        env = slot["env"]
        env.load_instances(**instances)
        net = self._rollout_net()
        self._set_padding_hint(net, "n_agents" in instances)
        sm, rec, n = self.rollout(net, slot, ss, "test", record=self.keep_greedy_record)
        trunc = self._check_flags(env, "run_test")
        self.last = dict(summary=sm, n_steps=n, rec=rec, truncated=trunc)
        sm = sm.cpu().numpy()
        out = {k: sm[:, 2 + i].copy() for i, k in enumerate(METRIC_KEYS)}
        out["reward"] = sm[:, 0].copy()
        return out

    # ------------------------------------------------------------------ runner.py:45-49 (greedy evaluation)
    def testing(self, agents_range=(10, 20), tasks_range=(20, 50), seed=None, seeds=None):
        """Greedy reward(s) of the local network on seeded instances; seed -> float, seeds -> numpy array.

        Defaults are the reference's AGENTS_RANGE / TASKS_RANGE (parameters.py:15-16, runner.py:45): with (lo, hi)
        ranges every seed draws its own sizes (env/task_env.py:58-65), so the seeds of one call form a ragged batch."""
        A = int(agents_range[1] if isinstance(agents_range, (tuple, list)) else agents_range)
        T = int(tasks_range[1] if isinstance(tasks_range, (tuple, list)) else tasks_range)
        ar = tuple(agents_range) if isinstance(agents_range, (tuple, list)) else int(agents_range)
        tr = tuple(tasks_range) if isinstance(tasks_range, (tuple, list)) else int(tasks_range)
        ss = [seed if seed is not None else 0] if seeds is None else list(seeds)
        out = []
        net = self._rollout_net()
        ragged = self._is_range(ar) or self._is_range(tr)
        self._set_padding_hint(net, ragged)
        for i in range(0, len(ss), self.B):
            chunk = ss[i:i + self.B]
            # a batch of exactly the seeds asked for (its env handle and graphs are cached per size like any other shape):
            # driver.py:245-250 asks for ONE seed per call, and padding that to n_envs copies cost a full-batch rollout each
            slot = self._slot(A, T, n_envs=len(chunk))
            env = slot["env"]
            inst = generate_batch_ranges(chunk, ar, tr)    # same draw order as TaskEnv(ar, tr, seed=s), sizes first
            if not ragged:
                inst.pop("n_agents"); inst.pop("n_tasks")  # uniform batch: shape-specialised kernels
            env.load_instances(**inst)
            cs = np.array([env_seeds(self.base_seed, int(s), 1)[0] for s in chunk], dtype=np.uint64)
            summary, _, _ = self.rollout(net, slot, cs, "greedy", record=False)
            self._check_flags(env, "testing")
            out.extend(summary[:, 0].cpu().numpy().tolist())
        return out[0] if seeds is None else np.array(out)

    def close(self):
        for slot in self._cache.values():
            slot["graphs"].clear()
            slot["env"].close()
        self._cache.clear()
