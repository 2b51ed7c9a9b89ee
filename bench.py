#!/usr/bin/env python3
"""bench.py -- env steps/sec of the batched coalition-formation + routing rollout (BASELINE.json metric).

One bench "step" = one pass of the hot path over one batch: every env of the batch plays 3 consecutive complete
episodes (auto-reset to the same instance, the decision counter keeps running) under the
uniform-random valid policy inside ONE launch of the persistent HIP kernel (dcm_rollout_random), with the observation
tensors + mask built and stored at every decision.  value = decisions taken by all envs on all
ranks / wall time, inputs (instances, seeds, state) resident in HBM before the timed region starts.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload at N=1 = BASELINE.json configs[1]: 4096 parallel envs, 20 agents / 50 tasks, random policy, HIP env
only.  N>1: the env batch is sharded (4096 envs per GPU, weak scaling), no data-path collective; one RCCL
all-gather of the per-env episode returns per pass (the analogue of ray.get in driver.py:129-130).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from dcmrta_amd.batched_env import BatchedTaskEnv  # noqa: E402
from dcmrta_amd.choice import env_seeds  # noqa: E402
from dcmrta_amd.dist import DistContext  # noqa: E402
from dcmrta_amd.instances import generate_batch  # noqa: E402
from dcmrta_amd.roofline import HBM_PEAK_BYTES_PER_S, algorithmic_bytes_per_step  # noqa: E402


def usable_cores():
    """Host threads this process may really run on: min(cpu_count, affinity mask, cgroup CPU quota).  (The GPU box shows
    256 hardware threads but a cgroup quota of 16 CPUs: 256 runnable threads get throttled to below the 16-thread rate.)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]      # cgroup v2
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())          # cgroup v1
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(inst, seeds, A, target_core_seconds=20.0):
    """The oracle (bit-parity C restatement of the reference) on the host cores: baseline, not target."""
    import oracle
    oracle.build()
    cores = usable_cores()
    B = len(seeds)
    # calibrate on a small slice, then size the sample to ~target_core_seconds of CPU work
    t0 = time.perf_counter()
    n0, *_ = oracle.batch_rollout(inst["depot"][:64], inst["task_xy"][:64], inst["req"][:64], inst["dur"][:64], seeds[:64], A,
                                  episodes=1, threads=1)
    rate1 = n0 / (time.perf_counter() - t0)
    per_episode = n0 / 64.0
    episodes = int(max(1, min(64, round(target_core_seconds * rate1 / (per_episode * B)))))
    t0 = time.perf_counter()
    n, *_ = oracle.batch_rollout(inst["depot"], inst["task_xy"], inst["req"], inst["dur"], seeds, A, episodes=episodes,
                                 threads=cores)
    dt = time.perf_counter() - t0
    return dict(value=n / dt, unit="steps/s", cores=cores, kind="port",
                sample=f"{B} envs x {episodes} episodes ({n} decisions) of the same instances/seeds, oracle C port, "
                       f"{cores} threads (= usable host CPUs: cpu_count {os.cpu_count()}, cgroup/affinity limit {cores}), "
                       f"one env per thread")


def lockstep_kernel_probe(A, T, dev, B=65536, n=24):
    """The lockstep kernel k_step really moves the algorithmic bytes (record in, record + observation out) once per
    decision: measured at a batch that fills the machine, HIP events around dcm_step only, device-side random policy."""
    env = BatchedTaskEnv(B, A, T, device=str(dev)).load_instances(**generate_batch(B, A, T, base_seed=0))
    obs = env.reset(env_seeds(0, 0, B))
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for i in range(n):
        act = torch.multinomial((~obs.mask).float(), 1).squeeze(1).int()
        ev[i][0].record()
        obs = env.step(act)
        ev[i][1].record()
    torch.cuda.synchronize(dev)
    ms = sorted(a.elapsed_time(b) for a, b in ev)[n // 2]
    Wb = algorithmic_bytes_per_step(A, T)
    return {"kernel": "k_step", "envs": B, "median_launch_ms": ms, "steps_per_s": B / ms * 1e3, "bound": "hbm",
            "achieved": B * Wb / ms / 1e6, "peak": HBM_PEAK_BYTES_PER_S / 1e9, "unit": "GB/s",
            "frac": B * Wb / ms / 1e6 / (HBM_PEAK_BYTES_PER_S / 1e9)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--agents", type=int, default=20)
    ap.add_argument("--tasks", type=int, default=50)
    ap.add_argument("--episodes", type=int, default=3,
                    help="consecutive episodes per env per pass (SURVEY.md §8d config 2: 3, auto-reset to the same instance)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-lockstep-probe", action="store_true")
    ap.add_argument("--no-obs", action="store_true", help="skip the observation stores (ablation, not the metric)")
    args = ap.parse_args()

    ctx = DistContext.from_env(expected_world=args.gpus)
    dev = ctx.device
    torch.cuda.set_device(dev)
    B, A, T = args.envs, args.agents, args.tasks
    first = ctx.rank * B
    inst = generate_batch(B, A, T, base_seed=0, first=first)
    seeds = env_seeds(0, first, B)
    env = BatchedTaskEnv(B, A, T, device=str(dev))
    env.load_instances(**inst)
    env.reset(seeds, observe=False)

    def one_pass():
        steps = env.rollout_random(episodes=args.episodes, write_obs=not args.no_obs)
        if ctx.world > 1:
            ctx.all_gather_returns(env.summary()[:, 0])
        return steps

    for _ in range(args.warmup):
        one_pass()
    K = args.steps
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
    counts, pending = [], []
    ctx.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(K):
        ev0[k].record()
        steps = env.rollout_random(episodes=args.episodes, write_obs=not args.no_obs)
        ev1[k].record()
        if ctx.world > 1:   # per-episode return exchange; overlaps with the next pass (nothing depends on it)
            pending.append(ctx.all_gather_returns(env.summary()[:, 0].contiguous(), async_op=True))
        counts.append(steps)
    for _, work in pending:
        if work is not None:
            work.wait()
    torch.cuda.synchronize(dev)
    ctx.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = ctx.max_over_ranks(elapsed)
    local_steps = int(torch.stack(counts).sum().item())
    total_steps = ctx.sum_over_ranks(local_steps)
    kern_s = sum(a.elapsed_time(b) for a, b in zip(ev0, ev1)) / 1e3

    flags = env.status()["flags"].cpu().numpy()
    assert (flags & 0x38).sum() == 0, "env error flags set"
    if ctx.rank != 0:
        ctx.shutdown()
        return

    Wb = algorithmic_bytes_per_step(A, T)
    achieved = local_steps * Wb / kern_s / 1e9
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            j = json.load(open(pmc))
            if j.get("workload") == f"{B}x{A}A{T}T":
                traffic = j.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    out = {
        "metric": "env_steps_per_sec", "value": total_steps / elapsed, "unit": "steps/s", "n_gpus": ctx.world, "steps": K,
        "warmup": args.warmup, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{B} envs/GPU x {A}A/{T}T random-policy rollout, HIP env only (BASELINE configs[1])",
                   "envs_per_gpu": B, "agents": A, "tasks": T, "episodes_per_step": args.episodes,
                   "decisions_per_step_per_gpu": local_steps / K, "sharding": f"env batch x{ctx.world}, no data-path collective"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_BYTES_PER_S / 1e9, "unit": "GB/s",
                     "frac": achieved * 1e9 / HBM_PEAK_BYTES_PER_S, "traffic": traffic, "kernel": "k_rollout_random",
                     "algorithmic_bytes_per_step": Wb, "avg_launch_ms": kern_s / K * 1e3},
    }
    if ctx.world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(inst, seeds, A)
    if ctx.world == 1 and not args.no_lockstep_probe:
        out["lockstep_kernel"] = lockstep_kernel_probe(A, T, dev)
    print(json.dumps(out), flush=True)
    ctx.shutdown()


if __name__ == "__main__":
    main()
