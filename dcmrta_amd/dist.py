"""Multi-GPU plumbing: one process per GPU, the env batch sharded by contiguous blocks.

Env instances are fully independent (each RLRunner of the reference is an isolated process,
runner.py:74-77), so stepping needs no collective.  The only exchange is an all-gather of the per-env
episode returns so that every rank holds the full return vector -- the analogue of ray.get on the
8 runner results (driver.py:129-130) and of the reward vectors fed to ttest_rel (driver.py:244-280).
torch.distributed backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used by the CPU tests.
"""
import datetime
import os
from dataclasses import dataclass

import torch
import torch.distributed as dist


def _raw_bits(x):
    """The values' bit patterns as int64 (reinterpreted, never converted): float64 / int64 as they are, 32-bit types widened."""
    x = x.contiguous()
    if x.dtype in (torch.float64, torch.int64):
        return x.view(torch.int64)
    if x.dtype in (torch.float32, torch.int32):
        return x.view(torch.int32).to(torch.int64)
    if x.dtype in (torch.float16, torch.bfloat16, torch.int16):
        return x.view(torch.int16).to(torch.int64)
    raise TypeError(f"verify_gather: unsupported dtype {x.dtype}")


class _TrimAfterWait:
    """Work handle of the uneven-shard gather: wait() = the collective's wait() + the trim of the padded blocks."""

    def __init__(self, work, finish):
        self._work, self._finish = work, finish

    def wait(self):
        self._work.wait()
        if self._finish is not None:
            self._finish()
            self._finish = None
        return True


def shard_range(n_total, rank, world):
    """Contiguous block [lo, hi) of rank; the first n_total % world ranks hold one extra env."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


@dataclass
class DistContext:
    rank: int
    world: int
    local_rank: int
    device: torch.device
    backend: str = ""

    @classmethod
    def from_env(cls, expected_world=None, backend=None, device=None, timeout_s=None):
        """timeout_s: limit of the process-group bring-up and of every collective (default: DCM_DIST_TIMEOUT, else 120 s) -- a rank
        that hangs makes the others fail with a reason instead of holding the job until the caller's own limit."""
        # before anything initialises HIP/HSA: the host driver of this pool only supports dmabuf IPC (without it RCCL fails
        # with hipIpcGetMemHandle: invalid argument)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        world = int(os.environ.get("WORLD_SIZE", "1"))
        rank = int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if expected_world is not None and expected_world != world:
            raise RuntimeError(f"--gpus {expected_world} but WORLD_SIZE={world}: either start the script plainly (it launches "
                               f"its own ranks, dcmrta_amd/launch.py) or use torch.distributed.run --nproc-per-node {expected_world}")
        if device is None:
            # DCM_FORCE_DEVICE lets several ranks share one GPU (functional smoke of the N>1 path on a 1-GPU box,
            # together with DCM_DIST_BACKEND=gloo; RCCL itself needs one device per rank)
            forced = os.environ.get("DCM_FORCE_DEVICE")
            idx = int(forced) if forced is not None else local
            # preflight BEFORE any other GPU call (device_count does not initialise the GPU): one device per local rank
            n_dev = torch.cuda.device_count()
            if n_dev > 0 and idx >= n_dev:
                raise RuntimeError(f"rank {rank} (local rank {local}) needs HIP device {idx} but this node shows {n_dev} device(s): "
                                   f"--gpus / WORLD_SIZE {world} is larger than the device count (set DCM_FORCE_DEVICE=<ordinal> "
                                   f"with DCM_DIST_BACKEND=gloo to share one GPU between ranks)")
            device = torch.device("cuda", idx) if torch.cuda.is_available() else torch.device("cpu")
        be = ""
        # DCM_DIST_FORCE_INIT=1: initialise the process group even for one rank (exercises the RCCL branch on a 1-GPU box)
        if world > 1 or os.environ.get("DCM_DIST_FORCE_INIT") == "1":
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            be = backend or os.environ.get("DCM_DIST_BACKEND") or ("nccl" if device.type == "cuda" else "gloo")
            if device.type == "cuda":
                torch.cuda.set_device(device)
            if not dist.is_initialized():
                kw = {"device_id": device} if (be == "nccl") else {}
                if timeout_s is None:
                    timeout_s = float(os.environ.get("DCM_DIST_TIMEOUT", "120"))
                dist.init_process_group(backend=be, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=float(timeout_s)),
                                        **kw)
        return cls(rank, world, local, device, be)

    def _coll_device(self):
        return torch.device("cpu") if self.backend == "gloo" else self.device

    def barrier(self):
        if self._active():
            dist.barrier()

    def _active(self):
        return self.world > 1 or (dist.is_available() and dist.is_initialized())

    @property
    def active(self):
        """True when a process group is up (N > 1, or one rank with DCM_DIST_FORCE_INIT=1): collectives really run."""
        return self._active()

    def all_gather_returns(self, local_returns, async_op=False, n_total=None, width=1):
        """local_returns [B_local * width] (any float dtype; `width` values per env, e.g. the returns of the `width` episodes
        of a pass) -> [n_total * width] on every rank, rank-major order (shard_range blocks of envs).
        async_op=True returns (out, work): the collective runs on RCCL's stream and the caller's stream keeps
        launching env kernels; call work.wait() before reading `out`.  n_total: number of envs over all ranks when the
        blocks are uneven (n_total % world != 0): the shorter blocks are padded for the collective and trimmed again."""
        if not self._active():
            return (local_returns, None) if async_op else local_returns
        local_returns = local_returns.contiguous()
        n_loc = local_returns.numel()
        even = n_total is None or n_total * width == self.world * n_loc
        if not even:
            # uneven blocks: every rank pads its block to ceil(n_total / world) for the collective; the gathered vector is
            # trimmed back to the shard_range blocks -- after the (still asynchronous) collective has finished
            pad = -(-n_total // self.world) * width
            buf = torch.zeros((pad,), dtype=local_returns.dtype, device=local_returns.device)
            buf[:n_loc] = local_returns
            sizes = [tuple(width * x for x in shard_range(n_total, r, self.world)) for r in range(self.world)]

            def trim(full, out=None):
                parts = [full[r * pad: r * pad + (hi - lo)] for r, (lo, hi) in enumerate(sizes)]
                return torch.cat(parts, out=out) if out is not None else torch.cat(parts)
            if async_op:
                full, work = self.all_gather_returns(buf, async_op=True)
                if work is None:
                    return trim(full), None
                out = torch.empty((n_total * width,), dtype=local_returns.dtype, device=local_returns.device)
                return out, _TrimAfterWait(work, lambda: trim(full, out))
            return trim(self.all_gather_returns(buf))
        if async_op and self.backend != "gloo":
            out = torch.empty((self.world * n_loc,), dtype=local_returns.dtype, device=local_returns.device)
            return out, dist.all_gather_into_tensor(out, local_returns, async_op=True)
        if async_op:
            return self.all_gather_returns(local_returns), None
        if self.backend == "gloo" and local_returns.is_cuda:      # gloo gathers through host memory
            parts = [torch.empty(local_returns.shape, dtype=local_returns.dtype) for _ in range(self.world)]
            dist.all_gather(parts, local_returns.cpu())
            return torch.cat(parts).to(local_returns.device)
        out = torch.empty((self.world * n_loc,), dtype=local_returns.dtype, device=local_returns.device)
        dist.all_gather_into_tensor(out, local_returns)
        return out

    def verify_gather(self, gathered, local_returns, first):
        """Raise -- on EVERY rank, after the same collectives, so that no rank is left waiting in one -- unless `gathered` is
        the rank-major concatenation of every rank's local vector on every rank: each rank finds its own block at
        [first, first + B_local) bit for bit, and all ranks hold the same vector (min == max over ranks of a
        position-weighted checksum of the raw bits)."""
        g, l = _raw_bits(gathered), _raw_bits(local_returns)
        own_ok = bool(torch.equal(g[first:first + l.numel()].cpu(), l.cpu()))
        same = True
        if self._active():
            # two independent position-weighted sums of the raw bit patterns, in wrapping int64 arithmetic (no float cast, no
            # dropped bits): vectors that differ anywhere disagree in at least one of them for all practical purposes
            idx = torch.arange(1, g.numel() + 1, dtype=torch.int64, device=g.device)
            h1 = (g * (2 * idx + 1)).sum()
            h2 = ((g ^ (g >> 29)) * (idx * idx + 0x9E3779B1)).sum()
            dev = self._coll_device()
            lo = torch.stack([h1, h2, torch.tensor(1 if own_ok else 0, dtype=torch.int64, device=g.device)]).to(dev)
            hi = lo.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            same = bool(torch.equal(lo[:2].cpu(), hi[:2].cpu()))
            all_own_ok = int(lo[2]) == 1
        else:
            all_own_ok = own_ok
        if not own_ok:
            raise RuntimeError(f"rank {self.rank}: gathered returns do not hold this rank's block at offset {first}")
        if not all_own_ok:
            raise RuntimeError(f"rank {self.rank}: another rank does not find its block in its gathered returns")
        if not same:
            raise RuntimeError("ranks hold different gathered return vectors")
        return True

    def max_over_ranks(self, x):
        if not self._active():
            return x
        t = torch.tensor([x], dtype=torch.float64, device=self._coll_device())
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, x):
        if not self._active():
            return x
        t = torch.tensor([x], dtype=torch.int64, device=self._coll_device())
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return int(t.item())

    def group_size(self):
        """Rank count as the process group (RCCL / gloo) itself reports it; None without a group."""
        return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else None

    def device_names(self):
        """Per-rank device description gathered over the group (rank order): "cuda:<ordinal> <marketing name>" -- rank 0 prints
        it next to process_group_ranks so that a line shows WHICH devices the N ranks really ran on."""
        me = str(self.device)
        if self.device.type == "cuda":
            me += " " + torch.cuda.get_device_name(self.device)
        if not self._active():
            return [me]
        names = [None] * dist.get_world_size()
        dist.all_gather_object(names, me)
        return names

    def shutdown(self):
        if self._active():
            dist.barrier()
            dist.destroy_process_group()
