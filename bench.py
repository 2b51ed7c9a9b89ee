#!/usr/bin/env python3
"""bench.py -- env steps/sec of the batched coalition-formation + routing rollout (BASELINE.json metric).

One bench "step" = one pass of the hot path over one batch.

--config 2 (default) = BASELINE.json configs[1]: 4096 parallel envs per GPU, 20 agents / 50 tasks; every env plays 3
    consecutive complete episodes (auto-reset to the same instance, the decision counter keeps
    running) under the uniform-random valid policy inside the persistent HIP kernel (dcm_rollout_random), with the
    observation tensors + mask built and stored at every decision.  N > 1 shards more envs (weak scaling).
--config 4 = BASELINE.json configs[3]: 65 536 envs of 50A/200T in total, sharded over the N ranks by contiguous blocks
    (strong scaling), one episode per env per pass.
--config 5 = BASELINE.json configs[4]: route replay (execute_by_route) of 100A/500T instances with dynamic task arrivals,
    65 536 envs in total sharded over the N ranks (strong scaling; BASELINE does not fix the count: chosen, like config 4's, so
    that the 8-GPU shard is still several rounds of resident waves); a step of this config = one agent_step call of the
    replay.  Preset routes are synthetic (the reference ships routes for 20A/50T only).  --visibility initial,batch,period,cap
    selects another dynamic-arrival schedule than the reference's hard-coded 20,20,10,100 (under which tasks 101..500 never
    appear): reported as a separate workload, never as the config-5 number.

value = steps taken by all envs on all ranks / wall time, inputs (instances, seeds, routes, state) resident in HBM before
the timed region starts.  There is no data-path collective; one RCCL all-gather of the per-env EPISODE returns per pass (every
episode of the pass: returns[B_local, episodes]) -- the analogue of ray.get in driver.py:129-130 -- issued asynchronously so
that it overlaps with the next pass.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|4|5] [--streams S]      (N > 1: starts its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...  (a launcher's ranks are used as is)

--streams S: the rank's env block is cut into S contiguous sub-batches, each with its own handle and HIP stream.  A pass
is then S launches; a launch lasts as long as its slowest env, and with several independent streams one sub-batch's tail
(few live waves) overlaps with the body of the others instead of idling the machine (DESIGN.md §6 "launch tail").  Default
(--streams 0): 4 / 2 / 1 sub-batches are timed in an untimed calibration before the warm-up and the fastest is used.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    # `python bench.py --gpus N` with no launcher around it: this process -- which has not imported torch and never touches
    # HIP -- starts the N ranks as children (python -m torch.distributed.run ... bench.py <same argv>), relays rank 0's JSON
    # line and exits with their code.  The reference's driver starts its own 8 actors the same way (driver.py:99).
    from dcmrta_amd.launch import maybe_self_launch
    maybe_self_launch(__file__)

# The sub-batches of a pass run on separate HIP streams; the runtime maps streams onto 4 hardware queues by default and
# kernels that share a queue serialise.  Must be set before the HIP runtime initialises (measured on MI355X, 4096 envs:
# 4 streams on 4 queues 5.3e8 steps/s, on 8 queues 9.5e8).  (tools/profile.sh exports it too: under rocprofv3 the profiler
# initialises HIP before this line runs.)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402
import torch  # noqa: E402

from dcmrta_amd import _lib  # noqa: E402
from dcmrta_amd.batched_env import BatchedTaskEnv  # noqa: E402
from dcmrta_amd.choice import env_seeds  # noqa: E402
from dcmrta_amd.dist import DistContext, shard_range  # noqa: E402
from dcmrta_amd.instances import generate_batch, synthetic_route_arrays  # noqa: E402
from dcmrta_amd.roofline import (HBM_PEAK_BYTES_PER_S, algorithmic_bytes_per_step, isa_e32_share, issue_roofline, load_counters,  # noqa: E402
                                 replay_kernel_name, rollout_kernel_name, staleness, step_kernel_name)

REFERENCE_VISIBILITY = (20, 20, 10, 100)      # env/task_env.py:567, :221
CONFIGS = {
    # name: envs, agents, tasks, episodes per pass, scaling, label, kernel
    # (kernel: "rollout" = dcm_rollout_random -- which kernel that is for the shape: roofline.rollout_kernel_name)
    "2": dict(envs=4096, agents=20, tasks=50, episodes=3, scaling="weak", label="BASELINE configs[1]", kernel="rollout"),
    "4": dict(envs=65536, agents=50, tasks=200, episodes=1, scaling="strong", label="BASELINE configs[3]", kernel="rollout"),
    # 65 536 envs in total (BASELINE does not fix the count; the same as config 4): the 8-GPU shard is 8192 envs = 2.3 rounds of the 14
    # waves a CU holds.  One round is latency-bound (2.99 ms for 3584 envs, 3.14 for 4096, 4.7 for 7168): at 8192 envs in total the
    # shard was one wave per SIMD and 8 GPUs would have been 3.5x one; at 32 768, 5.7x; here the measured per-shard rate predicts 6.7x
    "5": dict(envs=65536, agents=100, tasks=500, episodes=1, scaling="strong", label="BASELINE configs[4]", kernel="k_replay"),
}
AUTO_STREAM_CANDIDATES = (4, 2, 1)   # --streams 0: pick the fastest of these in an untimed calibration before the warm-up


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


def cpu_baseline(inst, seeds, A, target_core_seconds=12.0):
    """The oracle (bit-parity C restatement of the reference) on the host cores: baseline, not target.  Timed on all
    usable cores and on 8 threads (the reference runs NUM_META_AGENT = 8 one-core Ray actors, parameters.py:4,
    runner.py:74)."""
    import oracle
    oracle.build()
    cores = usable_cores()
    B = len(seeds)
    # calibrate on a small slice, then size each sample to ~target_core_seconds of CPU work
    t0 = time.perf_counter()
    n0, *_ = oracle.batch_rollout(inst["depot"][:64], inst["task_xy"][:64], inst["req"][:64], inst["dur"][:64], seeds[:64], A,
                                  episodes=1, threads=1)
    rate1 = n0 / (time.perf_counter() - t0)
    per_episode = n0 / 64.0
    episodes = int(max(1, min(64, round(target_core_seconds * rate1 / (per_episode * B)))))

    def timed(threads):
        t0 = time.perf_counter()
        n, *_ = oracle.batch_rollout(inst["depot"], inst["task_xy"], inst["req"], inst["dur"], seeds, A, episodes=episodes,
                                     threads=threads)
        return n, n / (time.perf_counter() - t0)
    n, rate_all = timed(cores)
    t8 = min(8, cores)
    _, rate8 = timed(t8)
    return dict(value=rate_all, unit="steps/s", cores=cores, kind="port",
                sample=f"{B} envs x {episodes} episodes ({n} decisions) of the same instances/seeds, oracle C port, "
                       f"{cores} threads (= usable host CPUs: cpu_count {os.cpu_count()}, cgroup/affinity limit {cores}), "
                       f"one env per thread; timed again on {t8} threads",
                at_8_threads=dict(value=rate8, unit="steps/s", cores=t8,
                                  note="mirror of the reference's NUM_META_AGENT = 8 one-core actors (runner.py:74)"),
                single_thread_rate=rate1)


def cpu_baseline_replay(inst, routes, route_len, A, visibility, target_core_seconds=12.0, reactive=True):
    """Config 5: the oracle's execute_by_route (reactive) on the host cores, one env per pthread at a time (orc_batch_replay), on a
    bounded sample of the same instances and routes."""
    import oracle
    oracle.build()
    cores = usable_cores()

    def run(n, threads):
        t0 = time.perf_counter()
        r = oracle.batch_replay(inst["depot"][:n], inst["task_xy"][:n], inst["req"][:n], inst["dur"][:n], routes[:n], route_len[:n],
                                reactive=reactive, visibility=visibility, threads=threads)
        return r["total"], time.perf_counter() - t0
    n0, dt0 = run(min(8, len(route_len)), 1)                 # calibrate on a few envs, one thread
    rate1 = n0 / dt0
    per_env = n0 / min(8, len(route_len))
    n_envs = int(max(cores, min(len(route_len), round(target_core_seconds * rate1 / per_env))))
    n, dt = run(n_envs, cores)
    return dict(value=n / dt, unit="steps/s", cores=cores, kind="port",
                sample=f"{n_envs} envs ({n} agent steps) of the same instances / routes, oracle C port of execute_by_route with "
                       f"{'dynamic visibility' if reactive else 'all tasks visible'}, {cores} threads, one env per thread",
                single_thread_rate=rate1)


def lockstep_kernel_probe(A, T, dev, B=65536, n=100, warm=5):
    """The lockstep kernel k_step really moves the algorithmic bytes (record in, record + observation out) once per
    decision: the HBM roofline of this path is quoted on it, at a batch that fills the machine, HIP events around
    dcm_step only, device-side random policy.  The window is the first ~100 decisions of an episode -- while every env is
    still active -- because a step's cost depends on where the episode is: 138-150 us during the first 20 decisions at 20A/50T
    (co-located groups decide several times per event), 162-170 us from decision 30 on (every decision ends its event).  The
    committed rocprofv3 profile (tools/profile_lockstep.sh B A T 100) covers the same window.  Runs right after the timed
    region, BEFORE the CPU baseline (after seconds of host-only work the GPU has clocked down).
    `frac` prices a launch with the ALGORITHMIC bytes W (the kernel skips clean sections, so this can exceed what
    HBM delivers); `traffic_frac` is the MEASURED HBM traffic of the committed profile of the same batch over the same time --
    the physical utilisation."""
    env = BatchedTaskEnv(B, A, T, device=str(dev)).load_instances(**generate_batch(B, A, T, base_seed=0))
    obs = env.reset(env_seeds(0, 0, B))
    for _ in range(warm):
        obs = env.step(torch.multinomial((~obs.mask).float(), 1).squeeze(1).int())
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for i in range(n):
        act = torch.multinomial((~obs.mask).float(), 1).squeeze(1).int()
        ev[i][0].record()
        obs = env.step(act)
        ev[i][1].record()
    torch.cuda.synchronize(dev)
    ms = sorted(a.elapsed_time(b) for a, b in ev)[n // 2]
    # what an event pair costs by itself (two marker packets with nothing between them): the part of `median_launch_ms` that is
    # not the kernel -- rocprofv3's kernel-trace average of the same window is shorter by about this much
    empty = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
    for a, b in empty:
        a.record()
        b.record()
    torch.cuda.synchronize(dev)
    bracket_us = sorted(a.elapsed_time(b) for a, b in empty)[len(empty) // 2] * 1e3
    Wb = algorithmic_bytes_per_step(A, T)
    c = load_counters(f"k_step:{B}x{A}A{T}T")
    out = {"kernel": step_kernel_name(A, T), "envs": B, "median_launch_ms": ms, "steps_per_s": B / ms * 1e3, "bound": "hbm",
           "achieved": B * Wb / ms / 1e6, "peak": HBM_PEAK_BYTES_PER_S / 1e9, "unit": "GB/s",
           "frac": B * Wb / ms / 1e6 / (HBM_PEAK_BYTES_PER_S / 1e9), "algorithmic_bytes_per_launch": B * Wb,
           "traffic": c.get("hbm_bytes_per_launch") if c else None,
           "traffic_frac": (c["hbm_bytes_per_launch"] / (ms * 1e-3) / HBM_PEAK_BYTES_PER_S) if c and c.get("hbm_bytes_per_launch") else None,
           "counters_source": c.get("source") if c else None,
           "rocprof_avg_launch_us": c.get("avg_launch_us") if c else None, "empty_event_pair_us": bracket_us,
           "stale": staleness(c, _lib.build_id()) if c else None}
    env.close()
    # ... and the BASELINE batch in the steady state of a collection loop (DCM_PARAM_AUTO_RESET, the first 150 steps untimed): every
    # launch then holds envs whose episode ends in it -- terminal metrics, restart, first event: the launch's slowest waves
    # (tools/lockstep_probe.py 4096 A T 200 steady measures the same)
    try:
        Bs = 4096
        env = BatchedTaskEnv(Bs, A, T, device=str(dev), auto_reset=True).load_instances(**generate_batch(Bs, A, T, base_seed=0))
        obs = env.reset(env_seeds(0, 0, Bs))
        for _ in range(150):
            obs = env.step(torch.multinomial((~obs.mask).float(), 1).squeeze(1).int())
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
        for a, b in ev:
            act = torch.multinomial((~obs.mask).float(), 1).squeeze(1).int()
            a.record()
            obs = env.step(act)
            b.record()
        torch.cuda.synchronize(dev)
        us = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
        out["steady_state_us"] = us[len(us) // 2]
        # (with deferred terminal metrics one dcm_step in 32 also launches k_terminal_flush: in the mean, not in the median)
        out["steady_state_mean_us"] = sum(us) / len(us)
        out["steady_state_envs"] = Bs
        # the kernel-trace average of the same loop from the committed profile (HIP events around dcm_step vary by 3 us from box to
        # box; the kernel's own duration says whether an episode end still stretches the launch)
        cs = load_counters(f"k_step_steady:{Bs}x{A}A{T}T")
        if cs and not staleness(cs, _lib.build_id()):
            out["steady_state_rocprof_avg_us"] = cs.get("avg_launch_us")
            out["terminal_flush_rocprof_avg_us"] = cs.get("flush_avg_us")
        c4 = load_counters(f"k_step:{Bs}x{A}A{T}T")                     # ... against the window in which no episode ends
        if c4 and not staleness(c4, _lib.build_id()):
            out["no_episode_end_rocprof_avg_us"] = c4.get("avg_launch_us")
        env.close()
    except Exception as ex:
        out["steady_state_error"] = f"{type(ex).__name__}: {ex}"[:200]
    return out


def _bits_equal(a, b):
    """Per-env equality of two [B, ...] arrays; float64 by bit pattern."""
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, (a.shape, b.shape, a.dtype, b.dtype)
    eq = (a.view(np.uint64) == b.view(np.uint64)) if a.dtype == np.float64 else (a == b)
    return eq.reshape(len(a), -1).all(axis=1)


def parity_rollout(sb, A, EP, cores):
    """Oracle parity of one sub-batch AT BENCH SIZE, after the timed region: the envs restart from their seeds, play one pass
    (EP episodes in the persistent kernel, observation stores on) and every env's decision total, every episode's return
    (dcm_set_return_log), last-episode reward, finished-task count and six perf metrics are compared bit for bit with the
    multi-threaded oracle on the same instances / seeds (worker.py:45-87,103-108).  Returns (envs, mismatching envs, oracle
    decisions, oracle seconds)."""
    import oracle
    oracle.build()
    sb.env.reset(sb.seeds, observe=False)
    steps = sb.env.rollout_random(episodes=EP, write_obs=True).cpu().numpy()
    sm = sb.env.summary().cpu().numpy()
    ring = sb.ring.cpu().numpy()
    t0 = time.perf_counter()
    ref = oracle.batch_rollout_full(sb.inst["depot"], sb.inst["task_xy"], sb.inst["req"], sb.inst["dur"], sb.seeds, A,
                                    episodes=EP, threads=cores)
    dt = time.perf_counter() - t0
    ok = _bits_equal(steps, ref["steps"]) & _bits_equal(ring, ref["returns"]) & _bits_equal(sm[:, 0], ref["reward"]) & \
        _bits_equal(sm[:, 1].astype(np.int32), ref["n_finished"]) & _bits_equal(sm[:, 2:8], ref["metrics"])
    return sb.B, int((~ok).sum()), ref["total"], dt


def parity_replay(sb, A, visibility, cores, max_envs=4096):
    """The same for route replay: the first `max_envs` envs of the sub-batch through the oracle's execute_by_route (same routes,
    same visibility schedule): agent_step totals, reward, finished-task count, six perf metrics, guard flag."""
    import oracle
    oracle.build()
    n = min(sb.B, max_envs)
    out = sb.env.execute_routes(sb.reactive, fields=())
    steps, flags, sm = out["steps"].cpu().numpy()[:n], out["flags"].cpu().numpy()[:n], out["summary"].cpu().numpy()[:n]
    t0 = time.perf_counter()
    ref = oracle.batch_replay(sb.inst["depot"][:n], sb.inst["task_xy"][:n], sb.inst["req"][:n], sb.inst["dur"][:n], sb.routes[:n],
                              sb.route_len[:n], reactive=sb.reactive, visibility=visibility, threads=cores)
    dt = time.perf_counter() - t0
    ok = _bits_equal(steps, ref["steps"]) & _bits_equal(sm[:, 0], ref["reward"]) & _bits_equal(sm[:, 1].astype(np.int32), ref["n_finished"]) & \
        _bits_equal(sm[:, 2:8], ref["metrics"]) & (((flags & 4) != 0) == (ref["status"] == 1)) & ((flags & 0x58) == 0)
    return n, int((~ok).sum()), ref["total"], dt


PARITY_FIELDS_ROLLOUT = ["decisions per env", "return of every episode of the pass", "last-episode reward", "finished tasks",
                         "success_rate, makespan, time_cost, waiting_time, travel_dist, efficiency"]
PARITY_FIELDS_REPLAY = ["agent_step calls per env", "reward", "finished tasks", "six perf metrics", "guard / error flags"]


def shard_roofline(kernel, A, T, units_per_pass, pass_s):
    """Issue roofline of an other_configs entry from the committed counters of that kernel (profiles/counters.json)."""
    c = load_counters(f"{kernel}:{A}A{T}T")
    if not c:
        return {"kernel": kernel, "frac": None, "note": f"no PMC profile committed for {kernel}:{A}A{T}T"}
    r = issue_roofline(c, units_per_pass, pass_s, unit="decision", e32_share=isa_e32_share(kernel))
    keep = {k: r.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "frac_hi", "valu_issue_frac", "salu_issue_frac")
            if r.get(k) is not None}
    t = c.get("hbm_bytes_per_decision")
    keep.update({"kernel": kernel, "traffic": t * units_per_pass if t is not None else None,
                 "hbm_frac": (t * units_per_pass / pass_s / HBM_PEAK_BYTES_PER_S) if t is not None else None,
                 "counters_source": c.get("source"), "stale": staleness(c, _lib.build_id())})
    return keep


def config3_probe(dev, B=4096, A=20, T=50, warm=30, n=150):
    """BASELINE configs[2]: the attention policy (the stand-in of dcmrta_amd/policy.py: the reference's architecture, stock
    PyTorch-ROCm ops, the reference's fp32 arithmetic) in the loop with the HIP env step -- worker.py:62-76 as one captured HIP
    graph per decision (policy forward, sampling, dcm_step with the fused next observation), envs auto-resetting so that the
    batch stays full.  Policy-bound and out of scope to tune (attention.py stays the consumer's): on the line because it is a
    BASELINE config and the only workload where the observations this env writes are consumed."""
    from dcmrta_amd.graph_rollout import GraphedRollout
    from dcmrta_amd.policy import AttentionNet
    torch.manual_seed(0)
    net = AttentionNet().to(dev).eval()
    net.assume_no_padding = True

    @torch.no_grad()
    def policy(ob):
        lp = net(ob.tasks, ob.agents, ob.mask)                       # Categorical(logp.exp()).sample() as an exponential race
        return torch.argmax(lp - torch.empty_like(lp).exponential_(1.0).log(), dim=1).to(torch.int32)
    env = BatchedTaskEnv(B, A, T, device=str(dev), auto_reset=True).load_instances(**generate_batch(B, A, T, base_seed=0))
    seeds = env_seeds(0, 0, B)
    g = GraphedRollout(env, policy, check_every=8)
    g.capture(seeds)
    obs = env.reset(seeds)
    for _ in range(warm):
        g.graph.replay()
    d0 = int(env.status()["decisions"].sum())
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(n):
        g.graph.replay()
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    dec = int(env.status()["decisions"].sum()) - d0
    # the env's share: HIP events around eager dcm_step calls on the same batch
    ev = []
    for _ in range(40):
        act = torch.argmax((~obs.mask).to(torch.int32), dim=1).int()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        obs = env.step(act)
        e1.record()
        ev.append((e0, e1))
    torch.cuda.synchronize(dev)
    env_ms = sorted(a.elapsed_time(b) for a, b in ev)[len(ev) // 2]
    out = {"workload": f"{B} envs x {A}A/{T}T, attention policy (stand-in, stock PyTorch-ROCm fp32 = the reference's policy arithmetic) "
                       f"sampled in the loop + HIP env step, one HIP graph per decision, auto-reset (BASELINE configs[2])",
           "steps_per_s_end_to_end": dec / wall, "value": dec / wall, "unit": "steps/s", "batched_steps": n,
           "ms_per_batched_step": wall / n * 1e3, "env_ms_per_batched_step": env_ms, "policy_share": 1 - env_ms / (wall / n * 1e3),
           "active_fraction": dec / (n * B), "policy_dtype": "fp32",
           "note": "policy-bound, out of scope: the env step is the HIP product, the policy is the consumer's (attention.py unchanged)"}
    env.close()
    return out


def other_config_shards(dev, visibility, passes=12):
    """BASELINE configs[3] and configs[4] on their per-GPU shard of an 8-GPU node (8192 envs x 50A/200T rollout; 8192 envs x
    100A/500T route replay with dynamic arrivals), one mid-size shape (4096 envs x 70A/130T) and BASELINE configs[2] (attention
    policy in the loop), timed inside the DEFAULT run so that whoever clocks `python bench.py` also clocks them: one warm pass,
    then `passes` passes back to back, inputs resident in HBM.  Each entry carries its issue roofline (committed
    counters of that kernel), a CPU baseline (the oracle on the same shard, all usable cores) and the oracle parity of the
    shard.  The full lines of these configs (sharding, streams) are `bench.py --config 4` / `--config 5`."""
    out = {}
    cores = usable_cores()
    side = [torch.cuda.Stream(device=dev) for _ in range(max(AUTO_STREAM_CANDIDATES))]
    main_stream = torch.cuda.current_stream(dev)

    def timed(c, B, eps, S, n_pass):
        """`n_pass` passes of S sub-batches back to back (one warm pass first); returns (seconds, steps, sub-batches)."""
        subs = [SubBatch(c, lo, hi - lo, dev, main_stream if S == 1 else side[k], visibility)
                for k, (lo, hi) in enumerate(shard_range(B, k, S) for k in range(S))]
        counts = []
        for rep in (1, n_pass):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(rep):
                for sb in subs:
                    with torch.cuda.stream(sb.stream):
                        counts.append(sb.run(eps, True)[0])
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
            if rep == 1:
                counts = []
        return dt, int(sum(int(x.sum().item()) for x in counts)), subs
    mid = dict(CONFIGS["2"], agents=70, tasks=130, episodes=3)     # a mid-size shape (env/task_env.py:57-65 draws sizes from ranges)
    for name, cfg, B, eps in (("config4_shard", CONFIGS["4"], 8192, 1), ("config5_shard", CONFIGS["5"], 8192, 1), ("midsize_70A130T", mid, 4096, 3)):
        try:
            A, T = cfg["agents"], cfg["tasks"]
            instances_block(A, T, 0, B)               # the whole shard once; the sub-batches of every trial slice it
            c = dict(cfg, kernel=rollout_kernel_name(A, T) if cfg["kernel"] == "rollout" else replay_kernel_name(A, T, 5, True, visibility[3]),
                     episodes=eps)
            # like the headline workload the shard is cut into sub-batches on separate HIP streams when that pays (a launch lasts as
            # long as its slowest env: with several independent streams one sub-batch's tail overlaps with the others' bodies):
            # 4 / 2 / 1 are tried with four passes each, the best is timed
            trial = {}
            for S in AUTO_STREAM_CANDIDATES:
                dt_, _, subs_ = timed(c, B, eps, S, 4)
                trial[S] = dt_ / 4
                for sb in subs_:
                    sb.env.close()
            S = max(c_ for c_, t_ in trial.items() if t_ <= 1.01 * min(trial.values()))     # (near-ties: the larger stream count)
            dt, n, subs = timed(c, B, eps, S, passes)
            what = f"the per-GPU shard of {cfg['label']} on 8 GPUs" if name != "midsize_70A130T" else \
                "random-policy rollout at a mid-size shape, 3 episodes per env per pass"
            out[name] = {"workload": f"{B} envs x {A}A/{T}T, {what}", "kernel": c["kernel"],
                         "value": n / dt, "unit": "steps/s", "ms_per_pass": dt / passes * 1e3, "passes": passes, "steps_per_pass": n / passes,
                         "streams": S, "stream_trial_ms_per_pass": {str(k): v * 1e3 for k, v in trial.items()},
                         "roofline": shard_roofline(c["kernel"], A, T, n / passes, dt / passes)}
            ne = bad = units = 0
            osec = 0.0
            for sb in subs:
                r = parity_replay(sb, A, visibility, cores, max_envs=4096 // S) if sb.replay else parity_rollout(sb, A, eps, cores)
                ne, bad, units, osec = ne + r[0], bad + r[1], units + r[2], osec + r[3]
            out[name]["parity"] = {"envs_checked": ne, "mismatches": bad, "fields": PARITY_FIELDS_REPLAY if subs[0].replay else PARITY_FIELDS_ROLLOUT}
            out[name]["cpu_baseline"] = {"value": units / osec, "unit": "steps/s", "cores": cores, "kind": "port",
                                         "sample": f"{ne} envs of this shard, one pass ({units} steps), oracle C port, {cores} threads"}
            for sb in subs:
                sb.env.close()
        except Exception as ex:      # (an auxiliary entry must never take the headline line down with it)
            out[name] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
        _INST.pop((cfg["agents"], cfg["tasks"]), None)
    try:
        out["config3"] = config3_probe(dev)
    except Exception as ex:
        out["config3"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
    return out


def _pick(d, *keys):
    return {k: d.get(k) for k in keys if isinstance(d, dict) and d.get(k) is not None}


def _sig(x, n=4):
    """Numbers of the summary rounded to n significant digits (the full values are in the objects before it)."""
    if isinstance(x, float):
        return float(f"{x:.{n}g}")
    if isinstance(x, dict):
        return {k: _sig(v, n) for k, v in x.items()}
    return x


def make_summary(out):
    """The few numbers a reader of the LAST 2000 characters of the line needs (the driver keeps only the tail): headline value and
    roofline fractions, bench-size parity, the lockstep kernel, and value / frac / parity / CPU baseline of every other config."""
    roof, par, lock = out.get("roofline") or {}, out.get("parity") or {}, out.get("lockstep_kernel") or {}
    sm = {"value": out["value"], "ms_per_step": out["ms_per_step"], "ms_min_max": [out.get("ms_per_step_min"), out.get("ms_per_step_max")],
          "kernel": roof.get("kernel"),
          "roofline": _pick(roof, "frac", "frac_hi", "salu_issue_frac", "stale"),
          "parity": _pick(par, "envs_checked", "mismatches"),
          "cpu_baseline": _pick(out.get("cpu_baseline") or {}, "value", "cores"),
          "lockstep_kernel": _pick(lock, "envs", "frac", "traffic_frac", "rocprof_avg_launch_us", "steady_state_us", "steady_state_rocprof_avg_us", "no_episode_end_rocprof_avg_us", "error")}
    for name, e in (out.get("other_configs") or {}).items():
        if not isinstance(e, dict):
            continue
        ent = _pick(e, "value", "error")
        r = e.get("roofline") or {}
        if r.get("frac") is not None:
            ent["frac"] = r["frac"]
        if isinstance(e.get("parity"), dict):
            ent["parity_mismatches"] = e["parity"].get("mismatches")
            ent["parity_envs"] = e["parity"].get("envs_checked")
        if isinstance(e.get("cpu_baseline"), dict):
            ent["cpu"] = e["cpu_baseline"].get("value")
        sm[name] = ent
    return _sig(sm)


LINE_TAIL = ("roofline", "parity", "cpu_baseline", "summary")


def order_line(out):
    """Key order of the JSON line: the contract scalars first, then the bulky objects (config, other_configs, lockstep_kernel), then
    roofline / parity / cpu_baseline, and the compact summary LAST so that it is what a 2000-character tail shows."""
    out["summary"] = make_summary(out)
    head = [k for k in out if k not in LINE_TAIL and not isinstance(out[k], dict)]
    bulky = [k for k in out if k not in LINE_TAIL and isinstance(out[k], dict)]
    return {k: out[k] for k in head + bulky + [k for k in LINE_TAIL if k in out]}


def aux_failures(out):
    """Auxiliary entries of the line that failed or disagree with the oracle (the headline parity exits 3 by itself)."""
    bad = []
    for name, e in (out.get("other_configs") or {}).items():
        if isinstance(e, dict) and e.get("error"):
            bad.append(f"other_configs.{name}: {e['error']}")
        elif isinstance(e, dict) and isinstance(e.get("parity"), dict) and e["parity"].get("mismatches"):
            bad.append(f"other_configs.{name}: {e['parity']['mismatches']} of {e['parity'].get('envs_checked')} envs differ from the oracle")
    for name in ("lockstep_kernel", "cpu_baseline"):
        if isinstance(out.get(name), dict) and out[name].get("error"):
            bad.append(f"{name}: {out[name]['error']}")
    return bad


_INST = {}      # (A, T) -> (first, B, arrays): the largest block of instances generated so far for the shape


def instances_block(A, T, first, B):
    """generate_batch(B, A, T, base_seed=0, first=first), served from the block already generated for this shape when it covers
    the range (the stream trials of a shard cut the same envs into 4 / 2 / 1 sub-batches: generated once, sliced)."""
    hit = _INST.get((A, T))
    if hit is not None and hit[0] <= first and first + B <= hit[0] + hit[1]:
        o = first - hit[0]
        return {k: v[o:o + B] for k, v in hit[2].items()}
    inst = generate_batch(B, A, T, base_seed=0, first=first)
    if hit is None or B >= hit[1]:
        _INST[(A, T)] = (first, B, inst)
    return inst


class SubBatch:
    """One contiguous block of the rank's envs: its own handle and (for more than one sub-batch) its own HIP stream."""

    def __init__(self, cfg, first, B, dev, stream, visibility, reactive=True):
        A, T = cfg["agents"], cfg["tasks"]
        self.first, self.B, self.stream, self.replay, self.reactive = first, B, stream, cfg["kernel"].startswith("k_replay"), reactive
        self.inst = instances_block(A, T, first, B)
        self.seeds = env_seeds(0, first, B)
        self.env = BatchedTaskEnv(B, A, T, device=str(dev))
        self.env.load_instances(**self.inst)
        if self.replay:
            # routes only over the tasks that can ever become visible under the schedule (the reference's cap hides the rest)
            self.routes, self.route_len = synthetic_route_arrays(self.inst["req"], A, max_task=min(T, visibility[3]) if reactive else None)
            self.env.set_visibility(*visibility)
            # synthetic routes send exactly req[t] <= 5 agents to task t, so a task never lists more than 5 members: 5 member slots
            # per task (an overflow would be flagged, checked after the run): 11.7 KB of LDS per 100A/500T env, 14 waves per CU
            self.env.load_route_arrays(self.routes, self.route_len, member_cap=5)
        else:
            self.ring = self.env.enable_return_log(cfg["episodes"])          # every episode's return of a pass
            self.env.reset(self.seeds, observe=False)
        self.counts, self.ev, self.warm = [], [], []

    def run(self, episodes, write_obs):
        """One pass of this sub-batch; returns (steps int64[B], returns f64[B, episodes])."""
        if self.replay:
            out = self.env.execute_routes(self.reactive, fields=())
            self.last_flags = out["flags"]                # (checked after the timed region: no extra launch for it)
            return out["steps"], out["summary"][:, :1]
        return self.env.rollout_random(episodes=episodes, write_obs=write_obs), self.ring


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="2")
    ap.add_argument("--envs", type=int, default=None, help="config 2: envs per GPU; configs 4 / 5: envs in total")
    ap.add_argument("--agents", type=int, default=None)
    ap.add_argument("--tasks", type=int, default=None)
    ap.add_argument("--episodes", type=int, default=None, help="consecutive episodes per env per pass (configs 2 / 4)")
    ap.add_argument("--streams", type=int, default=0,
                    help="sub-batches (HIP streams) per GPU; 0 = calibrate 4 / 2 / 1 before the warm-up and keep the fastest")
    ap.add_argument("--visibility", default=None,
                    help="config 5: initial,batch,period,cap of the dynamic-arrival schedule (default: the reference's 20,20,10,100); "
                         "'static' = no dynamic arrivals at all (execute_by_route with reactive_planning False, every task routed)")
    ap.add_argument("--dist-timeout", type=float, default=None,
                    help="N > 1: seconds a rank waits in process-group bring-up / a collective before it fails with a reason "
                         "(default: $DCM_DIST_TIMEOUT, else 120)")
    ap.add_argument("--launch-timeout", type=float, default=None,
                    help="N > 1, self-launched: wall-clock limit of the whole job (dcmrta_amd/launch.py; default 1500 s, 0 = none)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the oracle legs (cpu_baseline AND the bench-size parity check)")
    ap.add_argument("--no-lockstep-probe", action="store_true")
    ap.add_argument("--no-obs", action="store_true", help="skip the observation stores (ablation, not the metric)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="config 2, one GPU: do not time the per-GPU shards of configs 4 / 5 (other_configs) after the run")
    args = ap.parse_args()
    cfg = dict(CONFIGS[args.config])
    for k in ("envs", "agents", "tasks", "episodes"):
        if getattr(args, k) is not None:
            cfg[k] = getattr(args, k)
    A, T, EP = cfg["agents"], cfg["tasks"], cfg["episodes"]
    if cfg["kernel"] == "rollout":
        cfg["kernel"] = rollout_kernel_name(A, T)
    replay = cfg["kernel"] == "k_replay"
    static_replay = args.visibility == "static"
    visibility = (REFERENCE_VISIBILITY if (static_replay or not args.visibility) else tuple(int(x) for x in args.visibility.split(",")))
    if replay:
        EP = 1
        cfg["kernel"] = replay_kernel_name(A, T, 5, not static_replay, visibility[3])     # (5 member slots: see SubBatch)

    ctx = DistContext.from_env(expected_world=args.gpus, timeout_s=args.dist_timeout)
    device_names = ctx.device_names()
    dev = ctx.device
    torch.cuda.set_device(dev)
    if cfg["scaling"] == "weak":
        first, B = ctx.rank * cfg["envs"], cfg["envs"]
        n_total = cfg["envs"] * ctx.world
    else:
        n_total = cfg["envs"]
        first, hi = shard_range(n_total, ctx.rank, ctx.world)
        B = hi - first
    main_stream = torch.cuda.current_stream(dev)
    # one set of side streams for the calibration AND the run: which hardware queue a stream lands on is decided by the runtime
    # when the stream is created / first used, so a calibration on other stream objects would not describe the run
    side_streams = [torch.cuda.Stream(device=dev) for _ in range(max(max(AUTO_STREAM_CANDIDATES), args.streams))]

    def make_subs(S):
        out = []
        for k in range(S):
            lo, hi = shard_range(B, k, S)
            out.append(SubBatch(cfg, first + lo, hi - lo, dev, main_stream if S == 1 else side_streams[k], visibility,
                                reactive=not static_replay))
        torch.cuda.synchronize(dev)
        return out

    def calibrate(cand, passes=12):
        """Untimed: seconds per pass of `cand` sub-batches (one warm pass, then `passes` passes back to back)."""
        subs_c = make_subs(cand)
        for timed in (False, True):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(passes if timed else 1):
                for sb in subs_c:
                    with torch.cuda.stream(sb.stream):
                        sb.run(EP, not args.no_obs)
            torch.cuda.synchronize(dev)
            dt = (time.perf_counter() - t0) / passes
        for sb in subs_c:
            sb.env.close()
        return dt

    calibration = None
    if args.streams > 0:
        S = max(1, min(args.streams, B))
    elif replay:
        S = 1       # two resident waves per CU: a replay launch is many rounds of workgroups, there is no launch tail to hide
    else:
        # how many streams pay off depends on how the runtime maps them onto hardware queues (see GPU_MAX_HW_QUEUES above):
        # measure instead of assuming
        calibration = {c: calibrate(c) for c in AUTO_STREAM_CANDIDATES if c <= B}
        # (a near-tie goes to the larger stream count: its advantage -- overlapped launch tails -- grows with the number of passes
        #  in flight, which a short calibration understates)
        best = min(calibration.values())
        S = max(c for c, t in calibration.items() if t <= 1.01 * best)
    subs = make_subs(S)
    returns = [torch.empty((B, EP), dtype=torch.float64, device=dev) for _ in range(2)]   # double-buffered per-pass returns
    in_flight = [None, None]          # the gather still reading returns[i], if any

    def one_pass(k, timed):
        """Every sub-batch plays its episodes (one persistent launch each); N > 1: exchange the episode returns."""
        done = []
        free = None
        if ctx.active and in_flight[k & 1] is not None:
            # the gather of pass k-2 reads the buffer this pass writes: order the sub-batch streams after it (the collective
            # runs on RCCL's stream; work.wait() makes the main stream wait for it, the event hands that on)
            in_flight[k & 1].wait()
            free = torch.cuda.Event()
            free.record(main_stream)
        for sb in subs:
            with torch.cuda.stream(sb.stream):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                steps, rets = sb.run(EP, not args.no_obs)
                e1.record()
                if timed:
                    sb.counts.append(steps)
                    sb.ev.append((e0, e1))
                else:
                    sb.warm.append(steps)
                if ctx.active:
                    lo = sb.first - first
                    if free is not None:
                        sb.stream.wait_event(free)
                    returns[k & 1][lo:lo + sb.B].copy_(rets)         # the return of EVERY episode of the pass
                    d = torch.cuda.Event()
                    d.record()
                    done.append(d)
        if ctx.active:
            for d in done:
                main_stream.wait_event(d)
            # per-episode return exchange on RCCL's stream; the env kernels of the next pass do not wait for it
            out = ctx.all_gather_returns(returns[k & 1].view(-1), async_op=True, n_total=n_total, width=EP)
            in_flight[k & 1] = out[1]
            return out
        return None

    for w in range(args.warmup):
        g = one_pass(w, False)
        if g is not None and g[1] is not None:
            g[1].wait()
    torch.cuda.synchronize(dev)
    if ctx.active:
        # the gathered vector is the rank-major concatenation of the per-rank return vectors on EVERY rank
        local = (torch.cat([sb.env.summary()[:, :1] if sb.replay else sb.ring for sb in subs]).contiguous().view(-1))
        ctx.verify_gather(ctx.all_gather_returns(local, n_total=n_total, width=EP), local, first * EP)
    K = args.steps
    pending = []
    ctx.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(K):
        pending.append(one_pass(k, True))
    for g in pending:
        if g is not None and g[1] is not None:
            g[1].wait()
    torch.cuda.synchronize(dev)
    ctx.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = ctx.max_over_ranks(elapsed)
    local_steps = int(sum(int(torch.stack(sb.counts).sum().item()) for sb in subs))
    total_steps = ctx.sum_over_ranks(local_steps)
    launch_ms = [a.elapsed_time(b) for sb in subs for a, b in sb.ev]
    # spread of the timed passes: the passes of different sub-batches overlap (that is what the streams are for), so a pass is timed
    # by its completion -- end of pass k minus end of pass k-1 on the same stream, averaged over the sub-batches (HIP events)
    pass_ms = [float(np.mean([sb.ev[k - 1][1].elapsed_time(sb.ev[k][1]) for sb in subs])) for k in range(1, K)] or [elapsed / K * 1e3]
    warm_steps = int(sum(int(torch.stack(sb.warm).sum().item()) for sb in subs if sb.warm))

    for sb in subs:
        if replay:
            flags = sb.last_flags.cpu().numpy()
            assert (flags & 0x58).sum() == 0, "replay error flags set (bad action / member overflow / TypeError)"
        else:
            flags = sb.env.status()["flags"].cpu().numpy()
            assert (flags & 0x138).sum() == 0, "env error flags set"
    # (right after the timed region, before any host-only work lets the GPU clock down)
    lockstep = None
    if ctx.world == 1 and not args.no_lockstep_probe and not replay:
        try:
            lockstep = lockstep_kernel_probe(A, T, dev)
        except Exception as ex:      # (an auxiliary probe must never take the headline line down with it)
            lockstep = {"error": f"{type(ex).__name__}: {ex}"[:300]}
    # Oracle parity at bench size (part of the cpu_baseline leg, after the timed region): every env of this rank's block.
    parity = None
    if not args.no_cpu_baseline:
        cores = max(1, usable_cores() // ctx.world)
        checked = bad = 0
        for sb in subs:
            ne, nb, _, _ = parity_replay(sb, A, visibility, cores) if replay else parity_rollout(sb, A, EP, cores)
            checked, bad = checked + ne, bad + nb
        checked, bad = ctx.sum_over_ranks(checked), ctx.sum_over_ranks(bad)
        parity = {"envs_checked": checked, "mismatches": bad, "fields": PARITY_FIELDS_REPLAY if replay else PARITY_FIELDS_ROLLOUT,
                  "against": "oracle/ (C restatement of env/task_env.py + worker.py:45-87, pinned on reference-generated goldens), "
                             "same instances / seeds" + (" / routes" if replay else "") + ", one pass after the timed region",
                  "note": (f"first {min(subs[0].B, 4096)} envs of each rank's block" if replay else "every env of every rank's block")}
        if bad:
            print(f"bench.py: PARITY FAILURE: {bad} of {checked} envs differ from the oracle", file=sys.stderr, flush=True)
            ctx.shutdown()
            sys.exit(3)
    if ctx.rank != 0:
        ctx.shutdown()
        return

    step_s = elapsed / K
    dec_per_step = local_steps / K
    unit = "agent_step" if replay else "decision"
    Wb = algorithmic_bytes_per_step(A, T)
    vis_tag = "" if (not replay or visibility == REFERENCE_VISIBILITY) else ":vis" + "-".join(str(v) for v in visibility)
    if static_replay:
        vis_tag = ":static"
    c = load_counters(f"{cfg['kernel']}:{A}A{T}T{vis_tag}")      # per-step PMC averages measured by tools/profile.sh
    counters_shape = None
    if c is None and not replay:
        # no profile committed for this very shape: price it with the instruction counts of the profiled shape that runs the same
        # kernel (the per-decision counts of one kernel move by a few per cent with the shape: 406 vs 413 VALU at 20A/50T vs
        # 15A/35T), and say so in the line
        for key in {"k_rollout_fast": ("k_rollout_fast:20A50T",), "k_rollout_fast_g": ("k_rollout_fast_g:70A130T",)}.get(cfg["kernel"], ()):
            c = load_counters(key)
            if c is not None:
                counters_shape = key.split(":", 1)[1]
                break
    build = _lib.build_id()
    roof = {"kernel": cfg["kernel"], "avg_launch_ms": float(np.mean(launch_ms)), "launches_per_step": S,
            f"{unit}s_per_step": dec_per_step}
    if c:
        # Issue-bound roofline: the persistent kernels keep the env in LDS, so HBM is not what limits them (roofline.hbm
        # below); the binding resource is instruction issue.  Instruction counts per step come from the committed rocprofv3
        # PMC profile of this same command, priced per instruction class with the clocks of profiles/r03_calib; decisions and
        # times are live from this run (HIP events on the launch streams).
        roof.update(issue_roofline(c, dec_per_step, step_s, unit="decision", e32_share=isa_e32_share(cfg["kernel"])))
        traffic = c.get("hbm_bytes_per_decision")
        traffic = traffic * dec_per_step if traffic is not None and not args.no_obs else None
        roof.update({"traffic": traffic,
                     "hbm": {"achieved": (traffic / step_s / 1e9) if traffic else None, "peak": HBM_PEAK_BYTES_PER_S / 1e9,
                             "unit": "GB/s", "frac": (traffic / step_s / HBM_PEAK_BYTES_PER_S) if traffic else None,
                             "note": "measured HBM bytes (2*FETCH_SIZE + WRITE_SIZE) per pass / pass time"},
                     "counters_source": c.get("source"), "counters_build_id": c.get("build_id"), "build_id": build,
                     # (set when the counters were profiled on another shape of the same kernel: an estimate, not this shape's own)
                     "counters_shape": counters_shape,
                     # the counters describe the binary they were profiled on: a kernel edit without a re-profile shows here
                     "stale": staleness(c, build)})
    else:
        roof.update({"bound": "valu_issue", "achieved": None, "peak": None, "unit": "G SIMD-clocks/s (VALU pipe busy)",
                     "frac": None, "traffic": None, "build_id": build,
                     "note": f"no PMC profile committed for {cfg['kernel']}:{A}A{T}T{vis_tag} (profiles/counters.json)"})
    # SURVEY.md §8(d) prices a decision with W = 2S + O + 4 algorithmic bytes whatever the kernel really moves; for the
    # LDS-resident kernels that figure is NOT a utilisation (it exceeds the HBM peak) and is reported only for the record
    roof["w_scored"] = {"algorithmic_bytes_per_step": Wb, "equiv_GBps": dec_per_step * Wb / step_s / 1e9,
                        # measured HBM traffic as a fraction of the algorithmic bytes of the same steps (<< 1: state resident in LDS)
                        "traffic_over_algorithmic": (roof["traffic"] / (dec_per_step * Wb)) if roof.get("traffic") else None,
                        "note": "SURVEY §8(d) pricing; the record never leaves LDS, see roofline.hbm for real traffic"}
    what = (f"route replay with dynamic task arrivals (visibility schedule initial,batch,period,cap = {visibility}"
            + (", the reference's constants" if visibility == REFERENCE_VISIBILITY else ", GENERALISED: not the reference's constants")
            + f"), synthetic preset routes over tasks 1..{min(T, visibility[3])} (the tasks that can ever become visible)") if replay else "random-policy rollout"
    if static_replay:
        what = "route replay WITHOUT dynamic arrivals (reactive_planning False: not BASELINE configs[4]), synthetic preset routes over all tasks"
    out = {
        "metric": "env_steps_per_sec", "value": total_steps / elapsed, "unit": "steps/s", "n_gpus": ctx.world, "steps": K,
        "warmup": args.warmup, "ms_per_step": step_s * 1e3, "higher_is_better": True, "scaling": cfg["scaling"],
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": (f"{B} envs/GPU x {A}A/{T}T {what}, HIP env only ({cfg['label']})"
                                if cfg["scaling"] == "weak" else
                                f"{n_total} envs x {A}A/{T}T {what} sharded over {ctx.world} GPU(s), HIP env only "
                                f"({cfg['label']})"),
                   "step_definition": ("one agent_step call of execute_by_route (env/task_env.py:572-587)" if replay else
                                       "one leader decision (TaskEnv.step + updates + next observation)"),
                   "envs_per_gpu": B, "envs_total": n_total, "agents": A, "tasks": T, "episodes_per_step": EP,
                   "decisions_per_step_per_gpu": dec_per_step, "decisions_in_warmup_per_gpu": warm_steps, "streams_per_gpu": S,
                   "stream_calibration_ms_per_pass": ({str(k): v * 1e3 for k, v in calibration.items()} if calibration else None),
                   "visibility": ("static" if static_replay else list(visibility)) if replay else None,
                   "sharding": f"env batch x{ctx.world}, no data-path collective"
                               + (f", one async all-gather of the {EP} episode return(s) of every env per pass" if ctx.active else ""),
                   "dist_backend": ctx.backend or None, "world": ctx.world, "process_group_ranks": ctx.group_size(), "rank_devices": device_names,
                   "self_launched": os.environ.get("DCM_SELF_LAUNCHED") is not None},
        "ms_per_step_min": min(pass_ms), "ms_per_step_max": max(pass_ms),
        "roofline": roof,
        "parity": parity,
    }
    out["config"]["limits"] = {"members_per_task": 5, "members_per_task_wide_handle": 16, "A": 128, "T": 1023,
                               "note": "the reference's lists are unbounded (env/task_env.py:321-322); at its constants "
                                       "(COALITION_SIZE 5, 20A/50T..100A/500T) none of these limits is reached"}
    if lockstep is not None:
        out["lockstep_kernel"] = lockstep
    # (only in the full default run: the profiling / A-B tools pass --no-lockstep-probe --no-cpu-baseline and must see one kernel)
    if ctx.world == 1 and args.config == "2" and not (args.no_other_configs or args.no_lockstep_probe or args.no_cpu_baseline) and \
            all(getattr(args, k) is None for k in ("envs", "agents", "tasks")):
        for sb in subs:
            sb.env.close()
        out["other_configs"] = other_config_shards(dev, REFERENCE_VISIBILITY)
    if ctx.world == 1 and not args.no_cpu_baseline:
        sb = subs[0]
        inst = {k: np.concatenate([x.inst[k] for x in subs]) for k in sb.inst}
        try:
            if replay:
                # (route arrays of the sub-batches can differ in their cap: the baseline samples the first sub-batch's envs)
                out["cpu_baseline"] = cpu_baseline_replay(sb.inst, sb.routes, sb.route_len, A, visibility, reactive=not static_replay)
            else:
                out["cpu_baseline"] = cpu_baseline(inst, np.concatenate([x.seeds for x in subs]), A)
        except Exception as ex:
            out["cpu_baseline"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
    out = order_line(out)
    print(json.dumps(out), flush=True)
    ctx.shutdown()
    bad_aux = aux_failures(out)
    if bad_aux:
        # (the headline parity has already exited 3 above; an auxiliary entry that failed or disagreed with the oracle must not
        #  pass silently either)
        print("bench.py: AUXILIARY FAILURE: " + "; ".join(bad_aux), file=sys.stderr, flush=True)
        sys.exit(4)


if __name__ == "__main__":
    main()
