"""Multi-GPU plumbing: one process per GPU, the env batch sharded by contiguous blocks.

Env instances are fully independent (each RLRunner of the reference is an isolated process,
runner.py:74-77), so stepping needs no collective.  The only exchange is an all-gather of the per-env
episode returns so that every rank holds the full return vector -- the analogue of ray.get on the
8 runner results (driver.py:129-130) and of the reward vectors fed to ttest_rel (driver.py:244-280).
torch.distributed backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used by the CPU tests.
"""
import os
from dataclasses import dataclass

import torch
import torch.distributed as dist


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
    def from_env(cls, expected_world=None, backend=None, device=None):
        world = int(os.environ.get("WORLD_SIZE", "1"))
        rank = int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if expected_world is not None and expected_world != world:
            raise RuntimeError(f"--gpus {expected_world} but WORLD_SIZE={world}: launch with torch.distributed.run "
                               f"--nproc-per-node {expected_world}")
        if device is None:
            # DCM_FORCE_DEVICE lets several ranks share one GPU (functional smoke of the N>1 path on a 1-GPU box,
            # together with DCM_DIST_BACKEND=gloo; RCCL itself needs one device per rank)
            forced = os.environ.get("DCM_FORCE_DEVICE")
            idx = int(forced) if forced is not None else local
            device = torch.device("cuda", idx) if torch.cuda.is_available() else torch.device("cpu")
        be = ""
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            be = backend or os.environ.get("DCM_DIST_BACKEND") or ("nccl" if device.type == "cuda" else "gloo")
            if device.type == "cuda":
                torch.cuda.set_device(device)
            if not dist.is_initialized():
                kw = {"device_id": device} if (be == "nccl") else {}
                dist.init_process_group(backend=be, rank=rank, world_size=world, **kw)
        return cls(rank, world, local, device, be)

    def _coll_device(self):
        return torch.device("cpu") if self.backend == "gloo" else self.device

    def barrier(self):
        if self.world > 1:
            dist.barrier()

    def all_gather_returns(self, local_returns, async_op=False):
        """local_returns [B_local] (any float dtype) -> [world*B_local] on every rank, rank-major order.
        async_op=True returns (out, work): the collective runs on RCCL's stream and the caller's stream keeps
        launching env kernels; call work.wait() before reading `out`."""
        if self.world == 1:
            return (local_returns, None) if async_op else local_returns
        local_returns = local_returns.contiguous()
        if async_op and self.backend != "gloo":
            out = torch.empty((self.world * local_returns.numel(),), dtype=local_returns.dtype, device=local_returns.device)
            return out, dist.all_gather_into_tensor(out, local_returns, async_op=True)
        if async_op:
            return self.all_gather_returns(local_returns), None
        if self.backend == "gloo" and local_returns.is_cuda:      # gloo gathers through host memory
            parts = [torch.empty(local_returns.shape, dtype=local_returns.dtype) for _ in range(self.world)]
            dist.all_gather(parts, local_returns.cpu())
            return torch.cat(parts).to(local_returns.device)
        out = torch.empty((self.world * local_returns.numel(),), dtype=local_returns.dtype, device=local_returns.device)
        dist.all_gather_into_tensor(out, local_returns)
        return out

    def max_over_ranks(self, x):
        if self.world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=self._coll_device())
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, x):
        if self.world == 1:
            return x
        t = torch.tensor([x], dtype=torch.int64, device=self._coll_device())
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return int(t.item())

    def shutdown(self):
        if self.world > 1 and dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
