"""ShardedRunner -- the runner level of the reference as ONE PROCESS PER GPU.

The reference fans a training round out to NUM_META_AGENT = 8 Ray actors, each an isolated process holding a copy of the
policy weights and playing its own episodes (runner.py:74-77, driver.py:99,114-130), and collects the results with
ray.get (driver.py:129-130).  On one MI355X node the same shape is `torchrun --nproc-per-node 8`: rank r owns GPU r and one
BatchedRunner with its contiguous share of the round's env budget (dist.shard_range); the policy weights of rank 0 are
broadcast (the analogue of the state_dict that every job.remote call carries, driver.py:117); envs are independent, so a
round needs NO collective while it is played; afterwards ONE all-gather moves every env's terminal row -- reward + the six
perf metrics -- so that every rank holds the full return vector (what driver.py:139-140 averages and :244-280 feeds to
ttest_rel).  torch.distributed backend "nccl" is RCCL over xGMI on ROCm; "gloo" serves the CPU-rendezvous tests.

Where the experience goes is the learner's choice, and both are offered:
  * it STAYS RANK-LOCAL (default): every rank computes the REINFORCE loss on its own decisions and the gradients are
    all-reduced (`all_reduce_gradients`) -- the data-parallel learner, no experience traffic at all;
  * `gather_experience(jobResults, dst=0)` ships it to one learner rank, which is what the reference's single learner
    process (driver.py:135-188) sees.

    ctx = DistContext.from_env()                     # RANK / WORLD_SIZE / LOCAL_RANK from torchrun
    sr = ShardedRunner(total_envs=4096, ctx=ctx)
    sr.broadcast_weights()                           # rank 0 -> everybody
    jobResults, metrics, info = sr.job(episode, agents_num, tasks_num)     # metrics: means over ALL envs of the round
"""
import numpy as np
import torch
import torch.distributed as dist

from .dist import DistContext, shard_range
from .runner import METRIC_KEYS, BatchedRunner


class ShardedRunner:
    def __init__(self, total_envs, ctx=None, **runner_kwargs):
        self.ctx = ctx if ctx is not None else DistContext.from_env()
        self.total = int(total_envs)
        self.lo, self.hi = shard_range(self.total, self.ctx.rank, self.ctx.world)
        if self.hi <= self.lo:
            raise ValueError("fewer envs than ranks")
        # episode_stride = the whole budget, env_offset = this rank's shard: round e plays instances [e * total, (e + 1) * total)
        self.runner = BatchedRunner(metaAgentID=self.ctx.rank, n_envs=self.hi - self.lo, device=str(self.ctx.device),
                                    episode_stride=self.total, env_offset=self.lo, **runner_kwargs)
        self.last_summary = None

    # ------------------------------------------------------------------ weights
    def _bcast_module(self, module, src):
        """state_dict of `module` from rank `src` to every rank: one flat buffer per dtype (no conversion: fp64 parameters and
        integer / bool buffers travel bit for bit), the source rank's own tensors are not touched."""
        if not self.ctx.active:
            return
        groups = {}
        for p in module.state_dict().values():
            if torch.is_tensor(p):
                groups.setdefault(p.dtype, []).append(p.data)
        for dtype, params in groups.items():
            wire = torch.uint8 if dtype == torch.bool else dtype
            flat = torch.cat([p.reshape(-1).to(wire) for p in params])
            buf = flat.cpu() if self.ctx.backend == "gloo" else flat
            dist.broadcast(buf, src=src)
            if self.ctx.rank == src:
                continue
            flat = buf.to(params[0].device)
            o = 0
            for p in params:
                n = p.numel()
                p.copy_(flat[o:o + n].view_as(p).to(dtype))
                o += n

    def broadcast_weights(self, src=0):
        """Rank `src`'s policy and baseline weights on every rank (one flat buffer per network)."""
        r = self.runner
        self._bcast_module(r.localNetwork, src)
        self._bcast_module(r.localBaseline, src)
        r.set_weights(r.localNetwork.state_dict())          # refreshes a low-precision rollout shadow, if any
        return self

    def weights_checksum(self):
        """Sum of all policy parameters in float64 (a cheap cross-rank equality probe for tests)."""
        return float(sum(p.double().sum() for p in self.runner.localNetwork.state_dict().values() if torch.is_tensor(p)))

    # ------------------------------------------------------------------ one round
    def job(self, episodeNumber, agents_num, tasks_num, as_lists=False):
        """Every rank plays its shard of round `episodeNumber` with its (identical) local weights.  Returns
        (jobResults, metrics, info): jobResults = this rank's experience (runner.py:58-71 shape), metrics = means over ALL
        envs of the round (identical on every rank), info carries `returns` (float64[total], rank-major = env order)."""
        r = self.runner
        w, wb = r.localNetwork.state_dict(), r.localBaseline.state_dict()
        jobResults, _, info = r.job(w, wb, episodeNumber, agents_num, tasks_num, as_lists=as_lists)
        local = r.last["summary"].contiguous()                                    # [B_local, 8]
        full = self.ctx.all_gather_returns(local.view(-1), n_total=self.total, width=8).view(-1, 8)
        self.last_summary = full
        m = full[:, 2:8].mean(0).cpu().numpy()
        metrics = {k: float(m[i]) for i, k in enumerate(METRIC_KEYS)}
        info = dict(info, rank=self.ctx.rank, world=self.ctx.world, envs=(self.lo, self.hi), returns=full[:, 0].clone())
        return jobResults, metrics, info

    def testing(self, seeds, agents_range=(10, 20), tasks_range=(20, 50)):
        """Greedy rewards of the local network on `seeds` (runner.py:45-49, driver.py:244-252: 256 seeds per evaluation),
        the seeds sharded over the ranks, every rank returning the full vector in seed order."""
        seeds = list(seeds)
        lo, hi = shard_range(len(seeds), self.ctx.rank, self.ctx.world)
        mine = self.runner.testing(agents_range, tasks_range, seeds=seeds[lo:hi]) if hi > lo else np.zeros(0)
        t = torch.as_tensor(np.asarray(mine, dtype=np.float64), device=self.ctx.device)
        return self.ctx.all_gather_returns(t, n_total=len(seeds)).cpu().numpy()

    # ------------------------------------------------------------------ learner side
    def gather_experience(self, jobResults, dst=0):
        """Ship every rank's experience tensors (slots 0-6 of jobResults, stacked form) to rank `dst`: returns the
        rank-major concatenation there (the single-learner view of driver.py:135-164), None elsewhere."""
        if not self.ctx.active:
            return jobResults
        slots = [x if torch.is_tensor(x) else (torch.stack(x) if len(x) else None) for x in jobResults[:7]]
        n_loc = torch.tensor([0 if slots[0] is None else slots[0].shape[0]], dtype=torch.int64, device=self.ctx._coll_device())
        counts = [torch.zeros_like(n_loc) for _ in range(self.ctx.world)]
        dist.all_gather(counts, n_loc)
        counts = [int(c) for c in counts]
        n_max = max(counts)
        out = []
        for x in slots:
            if x is None:
                raise ValueError("gather_experience needs at least one decision on every rank")
            pad = torch.zeros((n_max,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
            pad[:x.shape[0]] = x
            send = pad.to(torch.uint8) if pad.dtype == torch.bool else pad
            if self.ctx.backend == "gloo":
                send = send.cpu()
            parts = [torch.empty_like(send) for _ in range(self.ctx.world)] if self.ctx.rank == dst else None
            dist.gather(send, parts, dst=dst)
            if self.ctx.rank == dst:
                cat = torch.cat([p[:c] for p, c in zip(parts, counts)]).to(x.device)
                out.append(cat.to(torch.bool) if x.dtype == torch.bool else cat)
        return (out + [[], []]) if self.ctx.rank == dst else None

    def all_reduce_gradients(self, module, n_local):
        """Data-parallel REINFORCE: gradients of a loss SUMMED over this rank's `n_local` decisions are summed over the
        ranks and divided by the global number of decisions -- the gradient of the mean loss over the whole round."""
        n = torch.tensor([float(n_local)], dtype=torch.float64, device=self.ctx._coll_device())
        if self.ctx.active:
            dist.all_reduce(n)
        params = list(module.parameters())
        for p in params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        if self.ctx.active and params:       # one collective over one flat buffer per dtype, not one per parameter
            groups = {}
            for p in params:
                groups.setdefault(p.grad.dtype, []).append(p.grad)
            for grads in groups.values():
                flat = torch.cat([g.reshape(-1) for g in grads])
                buf = flat.cpu() if self.ctx.backend == "gloo" else flat
                dist.all_reduce(buf)
                flat = buf.to(grads[0].device)
                o = 0
                for g in grads:
                    g.copy_(flat[o:o + g.numel()].view_as(g))
                    o += g.numel()
        for p in params:
            p.grad.div_(float(n))
        return int(n)

    def close(self):
        self.runner.close()
