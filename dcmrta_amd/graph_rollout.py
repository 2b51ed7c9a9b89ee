"""HIP-graph capture of the policy-in-the-loop step.

One decision of the batched loop -- policy forward on the observation tensors, action selection, the experience record,
dcm_step with the fused observation of the next decision (worker.py:62-83) -- is a fixed sequence of kernels on fixed
buffers, so it is captured once into a HIP graph (torch.cuda.CUDAGraph on ROCm) and replayed: no per-kernel launch cost,
no host work between the policy and the env, no host sync per decision.  The host only looks at `active` every
`check_every` replays to detect the end of the batch of episodes (steps on finished envs are no-ops on the device).
The policy itself stays stock PyTorch.

Experience (worker.py:77-83) is written inside the graph with one indexed store per field into [capacity, B, ...]
buffers at a device-side step counter, so recording needs no host involvement either.

Compaction (`buckets`): the env step costs microseconds, the policy forward milliseconds, and towards the end of a batch of
episodes most envs have finished.  With buckets = (1.0, 0.5, 0.25, ...) one graph is captured per bucket size; at every
check the host picks the smallest bucket that holds the envs still active, and that graph runs the policy only on those
rows (gathered by a static index buffer: the active envs first, finished ones as filler) and scatters the actions back.
"""
import threading

import torch

# Graph capture is a process-wide affair on HIP: one capture at a time.  (Actors on worker threads -- ray_compat.init(
# concurrent=True) -- draw a new batch shape every training round and re-capture while the others replay; captures run in
# capture_error_mode="thread_local", so the other threads' launches / allocations do not invalidate them.)
CAPTURE_LOCK = threading.RLock()


class GraphedRollout:
    def __init__(self, env, policy, check_every=8, warmup=3, record=False, capacity=None, buckets=None):
        """policy(obs) -> integer actions[B] computed with torch ops on env.device (no host syncs, no new persistent
        allocations); it is captured together with env.step.  record=True keeps what worker.py:77-83 appends per decision
        (agent / task observation, action, mask, deciding agent, plus the `active` flag of the env) for up to `capacity`
        batched steps.  buckets: fractions of B (descending, first must be 1.0) for which a compacted-policy graph is kept;
        `policy` must then accept any batch size (it is called as policy(obs_like) with the gathered rows; a policy with the
        attribute wants_rows = True is called as policy(obs_like, rows) with the env indices of those rows, None = all)."""
        self.env, self.policy, self.check_every = env, policy, int(check_every)
        B, A, T, dev = env.B, env.A, env.T, env.device
        self.action = torch.zeros((B,), dtype=torch.int32, device=dev)
        self._warmup = warmup
        self._epoch = None
        # whole check windows: run() advances `check_every` replays at a time, so the record must hold the window in which the
        # last episode ends (a capacity that is not a multiple used to raise "raise capacity" although nothing overflowed)
        cap = int(capacity) if capacity is not None else 6 * (A + T) + 64
        self.capacity = -(-cap // self.check_every) * self.check_every
        self.rec = None
        if record:
            S = self.capacity
            self.rec = dict(agents=torch.empty((S, B, A, 6), dtype=torch.float32, device=dev),
                            tasks=torch.empty((S, B, T + 1, 5), dtype=torch.float32, device=dev),
                            mask=torch.empty((S, B, T + 1), dtype=torch.bool, device=dev),
                            action=torch.zeros((S, B), dtype=torch.int64, device=dev),
                            leader=torch.zeros((S, B), dtype=torch.int64, device=dev),
                            active=torch.zeros((S, B), dtype=torch.bool, device=dev))
            self.slot = torch.zeros((1,), dtype=torch.int64, device=dev)      # device-side step counter
        buckets = tuple(buckets) if buckets else (1.0,)
        if float(buckets[0]) != 1.0:
            raise ValueError("buckets must start with 1.0 (the full batch)")
        sizes = [B]
        for f in buckets[1:]:
            n = max(1, int(round(B * float(f))))
            if n < sizes[-1]:
                sizes.append(n)
        self.sizes = sizes                                                     # descending; sizes[0] == B
        self.idx = {n: torch.arange(n, dtype=torch.int64, device=dev) for n in sizes[1:]}   # static gather indices
        self.graphs = {}
        self.bucket_steps = {n: 0 for n in sizes}

    @property
    def graph(self):
        """The full-batch graph (None before capture)."""
        return self.graphs.get(self.env.B)

    def _one_step(self, obs, n):
        if n == self.env.B:
            a = self.policy(obs)
            self.action.copy_(a)
        else:
            # policy on the rows named by the bucket's index buffer only; the other envs (finished: their action is ignored by
            # the device) keep whatever self.action holds
            idx = self.idx[n]
            sub = type(obs)(obs.agents.index_select(0, idx), obs.tasks.index_select(0, idx), obs.mask.index_select(0, idx),
                            obs.leader.index_select(0, idx), obs.active.index_select(0, idx))
            a = self.policy(sub, idx) if getattr(self.policy, "wants_rows", False) else self.policy(sub)
            self.action.index_copy_(0, idx, a.to(torch.int32))
        if self.rec is not None:
            r, i = self.rec, self.slot
            r["agents"].index_copy_(0, i, obs.agents.unsqueeze(0))
            r["tasks"].index_copy_(0, i, obs.tasks.unsqueeze(0))
            r["mask"].index_copy_(0, i, obs.mask.unsqueeze(0))
            r["action"].index_copy_(0, i, self.action.to(torch.int64).unsqueeze(0))
            r["leader"].index_copy_(0, i, obs.leader.to(torch.int64).unsqueeze(0))
            r["active"].index_copy_(0, i, obs.active.unsqueeze(0))
            self.slot.add_(1).clamp_(max=self.capacity - 1)
        return self.env.step(self.action)

    def capture(self, seeds):
        with CAPTURE_LOCK:
            return self._capture(seeds)

    def _capture(self, seeds):
        env = self.env
        self.graphs = {}
        for n in self.sizes:
            obs = env.reset(seeds)
            s = torch.cuda.Stream(device=env.device)
            s.wait_stream(torch.cuda.current_stream(env.device))
            with torch.cuda.stream(s):                 # warm-up on a side stream, as torch's capture recipe requires
                for _ in range(self._warmup):
                    obs = self._one_step(obs, n)
            torch.cuda.current_stream(env.device).wait_stream(s)
            # (an auto-resetting handle defers the terminal metrics of its eager steps -- the warm-up above -- and refuses a capture
            #  while such summaries are waiting: dcm_summary computes them)
            env.summary()
            torch.cuda.synchronize(env.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                self._one_step(obs, n)
            self.graphs[n] = g
        # the captured dcm_step bakes in the handle's kernel arguments (per-env sizes pointer, route-log pointers, shape
        # instantiation): a later change of any of them must re-capture
        self._epoch = env.graph_epoch
        return self

    def _pick_bucket(self, obs):
        """Smallest bucket that holds the active envs; refreshes its index buffer (active envs first).  One host sync."""
        n_act = int(obs.active.sum())
        if n_act == 0:
            return 0, 0
        n = self.sizes[0]
        for cand in self.sizes[1:]:
            if cand >= n_act:
                n = cand
        if n != self.env.B:
            order = torch.argsort((~obs.active).to(torch.int8), stable=True)   # active envs first, finished ones as filler
            self.idx[n].copy_(order[:n])
        return n, n_act

    @torch.no_grad()
    def run(self, seeds, max_steps=None):
        """Play the batch of episodes; returns (summary[B,8], batched_steps).  With record=True the experience of batched
        step s is self.rec[...][s] (valid where rec["active"][s])."""
        env = self.env
        if not self.graphs or self._epoch != env.graph_epoch:
            self.capture(seeds)
        obs = env.reset(seeds)
        if self.rec is not None:
            self.slot.zero_()
        self.bucket_steps = {n: 0 for n in self.sizes}
        limit = max_steps if max_steps is not None else (self.capacity if self.rec is not None else 1 << 30)
        n, bucket = 0, env.B
        while True:
            g = self.graphs[bucket]
            # (replays are only enqueued here -- microseconds of host time; they must not interleave with another thread's
            #  capture: the sampler draws from torch's default CUDA generator, whose graph bookkeeping is per process)
            with CAPTURE_LOCK:
                for _ in range(self.check_every):
                    g.replay()
            n += self.check_every
            self.bucket_steps[bucket] += self.check_every
            # obs tensors are the env's static output buffers; one sync per check
            if len(self.sizes) == 1:
                if not bool(obs.active.any()):
                    break
            else:
                bucket, n_act = self._pick_bucket(obs)
                if n_act == 0:
                    break
            if n >= limit:
                raise RuntimeError(f"episodes still running after {n} batched steps (capacity / max_steps {limit}): "
                                   f"raise `capacity`")
        if self.rec is not None and n > self.capacity:
            raise RuntimeError(f"{n} batched steps recorded into {self.capacity} slots: raise `capacity`")
        self.steps = n
        return env.summary(), n
