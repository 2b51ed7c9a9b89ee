#!/usr/bin/env python3
"""bench_configs.py -- the BASELINE.json configs other than the headline one (bench.py measures configs[1]).

    python bench_configs.py [--config 1|2l|3|4|5|all]

Prints one JSON line per config (1 GPU; config 4's sharded variant is bench.py --gpus N --envs 8192 --agents 50 --tasks 200):
  1   1 env, 20A/50T test-set instance 0: oracle (CPU, 1 thread) vs HIP lockstep, per-step latency
  2l  config 2 through the LOCKSTEP API (dcm_step per decision, uniform-random valid action chosen by a torch op)
  2g  the same loop as eager launches vs ONE HIP graph per decision
  3   4096 envs 20A/50T, attention policy (stock PyTorch-ROCm) + HIP env step: eager fp32 / HIP graph fp32 / HIP graph with a
      bf16 and an fp16 shadow of the net, policy : env time split, end-to-end steps/s
  7   runner level: BatchedRunner.job (sampled episode + greedy twin + experience) at 256 / 4096 envs
  6   runtime-shape path: the persistent kernel and k_step at 4096 x 15A/35T, 4096 x 20A/49T and a ragged (10-20) x (20-50)
      batch next to the exact 20A/50T instantiation
  4   8192 envs/GPU 50A/200T random-policy rollout (the per-GPU shard of 65536 envs over 8 GPUs)
  5   100A/500T route replay, synthetic routes, with/without dynamic visibility
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
from dcmrta_amd.instances import generate_batch, load_instances_npz  # noqa: E402
from dcmrta_amd.roofline import HBM_PEAK_BYTES_PER_S, algorithmic_bytes_per_step  # noqa: E402

DEV = "cuda:0"


def sync():
    torch.cuda.synchronize()


def random_valid_action(mask):
    """Uniform over unmasked actions, on the device (a torch op standing in for a policy)."""
    p = (~mask).to(torch.float32)
    return torch.multinomial(p, 1).squeeze(1).to(torch.int32)


def lockstep_episode(env, seeds, policy):
    obs = env.reset(seeds)
    n = 0
    t_env = 0.0
    while True:
        # one host sync per batched step (all envs finished?) -- part of the lockstep protocol
        if not bool(obs.active.any()):
            break
        n += int(obs.active.sum())
        a = policy(obs)
        sync(); t0 = time.perf_counter()
        obs = env.step(a)
        sync(); t_env += time.perf_counter() - t0
    return n, t_env


def config1():
    import oracle
    inst, A = load_instances_npz(os.path.join(ROOT, "tests", "golden", "instances_20A50T.npz"))
    one = {k: v[:1] for k, v in inst.items()}
    seeds = env_seeds(0, 0, 1)
    o = oracle.OracleEnv(A, 50).load(one["depot"][0], one["task_xy"][0], one["req"][0], one["dur"][0])
    t0 = time.perf_counter(); reps = 200; n = 0
    for r in range(reps):
        o.clear_decisions()
        n += o.rollout(int(seeds[0]), 0, oracle.POLICY_RANDOM, record=False)["n_steps"]
    cpu = n / (time.perf_counter() - t0)
    env = BatchedTaskEnv(1, A, 50, device=DEV).load_instances(**one)
    lockstep_episode(env, seeds, lambda ob: random_valid_action(ob.mask))
    sync(); t0 = time.perf_counter()
    n2, t_env = lockstep_episode(env, seeds, lambda ob: random_valid_action(ob.mask))
    wall = time.perf_counter() - t0
    return dict(config=1, workload="1 env 20A/50T test-set instance 0", oracle_cpu_steps_per_s=cpu,
                hip_lockstep_steps_per_s=n2 / wall, hip_env_only_us_per_step=t_env / n2 * 1e6,
                reference_python_steps_per_s="~820 (BASELINE.md, survey container)")


def config2_lockstep(B=4096, A=20, T=50):
    inst = generate_batch(B, A, T, 0)
    env = BatchedTaskEnv(B, A, T, device=DEV).load_instances(**inst)
    seeds = env_seeds(0, 0, B)
    lockstep_episode(env, seeds, lambda ob: random_valid_action(ob.mask))
    sync(); t0 = time.perf_counter()
    n, t_env = lockstep_episode(env, seeds, lambda ob: random_valid_action(ob.mask))
    wall = time.perf_counter() - t0
    W = algorithmic_bytes_per_step(A, T)
    return dict(config="2-lockstep", workload=f"{B} envs {A}A/{T}T, dcm_step per decision + torch.multinomial policy",
                steps_per_s_end_to_end=n / wall, steps_per_s_env_only=n / t_env,
                env_only_hbm_frac=n / t_env * W / HBM_PEAK_BYTES_PER_S)


def config2_graph(A=20, T=50):
    """Same lockstep loop, eager launches vs one HIP graph per decision (dcmrta_amd/graph_rollout.py)."""
    from dcmrta_amd.graph_rollout import GraphedRollout
    out = {}
    for B in (256, 4096):
        inst = generate_batch(B, A, T, 0)
        env = BatchedTaskEnv(B, A, T, device=DEV).load_instances(**inst)
        seeds = env_seeds(0, 0, B)
        pol = lambda ob: torch.multinomial((~ob.mask).float(), 1).squeeze(1)
        obs = env.reset(seeds)
        sync(); t0 = time.perf_counter(); k = 0
        while k < 400:
            for _ in range(8):
                obs = env.step(pol(obs).int()); k += 1
            if not bool(obs.active.any()):
                break
        sync(); eager = (time.perf_counter() - t0) / k
        g = GraphedRollout(env, pol, check_every=8).capture(seeds)
        sync(); t0 = time.perf_counter()
        _, n = g.run(seeds)
        sync(); graphed = (time.perf_counter() - t0) / n
        out[f"B{B}"] = dict(eager_us_per_batched_step=eager * 1e6, graph_us_per_batched_step=graphed * 1e6)
    return dict(config="2-graph", workload=f"{A}A/{T}T lockstep loop (torch.multinomial policy + dcm_step), eager vs HIP graph", **out)


def config3(B=4096, A=20, T=50):
    """BASELINE configs[2]: attention policy in the loop, 4096 envs x 20A/50T.

    The reference's policy arithmetic is fp32 (attention.py), so the fp32 rows are THE config-3 numbers; the bf16 / fp16 rows run
    a low-precision shadow of the same weights and are opt-in results, listed with their measured action agreement against the
    fp32 net on the same observations.  Modes (each for fp32 and fp16):
      graph           one sampled episode per env, one captured HIP graph per decision, host looks at `active` every 8 steps
      tuned           + GEMMs picked by PyTorch's TunableOp (stock torch.cuda.tunable) -- all later rows use the tuned GEMMs
      3ep_auto_reset  3 consecutive episodes per env (SURVEY.md §8d "same envs" as config 2), DCM_PARAM_AUTO_RESET
      compacted       policy only on the envs still active (one graph per bucket size), 1 and 3 episodes per env
      steady_state    every env restarts forever: the batch is always full, 400 batched steps timed"""
    from dcmrta_amd.graph_rollout import GraphedRollout
    from dcmrta_amd.policy import AttentionNet
    torch.manual_seed(0)
    net = AttentionNet().to(DEV).eval()
    net.assume_no_padding = True
    inst = generate_batch(B, A, T, 0)
    env = BatchedTaskEnv(B, A, T, device=DEV).load_instances(**inst)
    seeds = env_seeds(0, 0, B)
    rows = {}

    def sampler(m):
        @torch.no_grad()
        def policy(ob):
            lp = m(ob.tasks, ob.agents, ob.mask)                       # Categorical(logp.exp()).sample() as an exponential race
            return torch.argmax(lp - torch.empty_like(lp).exponential_(1.0).log(), dim=1).to(torch.int32)
        return policy
    nets = {"fp32": net, "bf16": net.rollout_copy(torch.bfloat16), "fp16": net.rollout_copy(torch.float16)}
    # eager, fp32, one host sync per decision (round 1's loop)
    pol = sampler(net)
    lockstep_episode(env, seeds, pol)
    sync(); t0 = time.perf_counter()
    n, t_env = lockstep_episode(env, seeds, pol)
    wall = time.perf_counter() - t0
    rows["eager_fp32"] = dict(precision="fp32", steps_per_s_end_to_end=n / wall, steps_per_s_env_only=n / t_env,
                              policy_share=1 - t_env / wall, mean_reward=float(env.summary()[:, 0].mean()))
    # env-only cost of one batched dcm_step at this batch (for the split of the graphed variants)
    obs = env.reset(seeds)
    act = torch.zeros(B, dtype=torch.int32, device=DEV)
    ev = []
    for _ in range(60):
        act.copy_(torch.argmax((~obs.mask).to(torch.int32), dim=1))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); obs = env.step(act); e1.record()
        ev.append((e0, e1))
    sync()
    env_ms = sorted(a.elapsed_time(b) for a, b in ev)[len(ev) // 2]
    # action agreement of the low-precision shadows with the fp32 net on real mid-episode observations
    with torch.no_grad():
        obs = env.reset(seeds)
        agree = {k: [] for k in ("bf16", "fp16")}
        tv = {k: [] for k in ("bf16", "fp16")}
        for step in range(60):
            lp32 = net(obs.tasks, obs.agents, obs.mask)
            if step % 10 == 0:
                for k in agree:
                    lpk = nets[k](obs.tasks, obs.agents, obs.mask).float()
                    a = obs.active
                    agree[k].append(float((lpk.argmax(1) == lp32.argmax(1))[a].float().mean()))
                    tv[k].append(float(0.5 * (lpk.exp() - lp32.exp()).abs().sum(1)[a].mean()))
            obs = env.step(torch.argmax(lp32 - torch.empty_like(lp32).exponential_(1.0).log(), dim=1).int())
    agreement = {k: dict(greedy_action_agreement=float(np.mean(agree[k])), mean_total_variation=float(np.mean(tv[k])),
                         note="vs the fp32 net on the same observations (6 decision points of a sampled fp32 rollout, all active envs)")
                 for k in agree}

    def timed_run(e, m, **gkw):
        g = GraphedRollout(e, sampler(m), **gkw)
        g.run(seeds)
        sync(); t0 = time.perf_counter()
        summary, batched = g.run(seeds)
        sync(); wall = time.perf_counter() - t0
        dec = int(e.status()["decisions"].sum())
        return dict(steps_per_s_end_to_end=dec / wall, batched_steps=batched, ms_per_batched_step=wall / batched * 1e3,
                    active_fraction=dec / (batched * B), mean_reward=float(summary[:, 0].mean())), g
    for k in ("fp32", "bf16", "fp16"):
        r, _ = timed_run(env, nets[k], check_every=8)
        rows[f"graph_{k}"] = dict(precision=k, **r, env_ms_per_batched_step=env_ms,
                                  policy_share=1 - env_ms / r["ms_per_batched_step"])
    # PyTorch's TunableOp (torch.cuda.tunable: stock PyTorch, picks the fastest rocBLAS / hipBLASLt solution per GEMM shape at
    # first use) -- the remaining rows run with the tuned GEMMs
    import torch.cuda.tunable as tunable
    tunable.enable(True)
    tunable.tuning_enable(True)
    tunable.set_max_tuning_duration(30)
    tunable.set_filename(os.path.join(os.environ.get("TMPDIR", "/tmp"), "dcmrta_tunableop.csv"))   # (its results file)
    BUCKETS = (1.0, 0.75, 0.5, 0.375, 0.25, 0.125, 0.0625)
    for k in ("fp32", "fp16"):
        m = nets[k]
        t0 = time.perf_counter()
        with torch.no_grad():
            obs = env.reset(seeds)
            for _ in range(2):
                m(obs.tasks, obs.agents, obs.mask)
            for frac in BUCKETS[1:]:                      # the compacted graphs call the net at the bucket sizes too
                nb = max(1, int(round(B * frac)))
                m(obs.tasks[:nb], obs.agents[:nb], obs.mask[:nb])
        sync()
        tune_s = time.perf_counter() - t0
        tunable.tuning_enable(False)                  # (keep using the tuned solutions; never tune inside a graph capture)
        r, _ = timed_run(env, m, check_every=8)
        rows[f"tuned_{k}"] = dict(precision=k, **r, gemm_tuning_seconds=tune_s)
        env3 = BatchedTaskEnv(B, A, T, device=DEV, auto_reset=True, auto_reset_episodes=3).load_instances(**inst)
        r, _ = timed_run(env3, m, check_every=8)
        rows[f"tuned_{k}_3ep_auto_reset"] = dict(precision=k, **r, episodes=int(env3.episodes().sum()))
        env3.close()
        for key, kw in ((f"tuned_{k}_compacted", {}), (f"tuned_{k}_3ep_auto_reset_compacted", dict(auto_reset=True, auto_reset_episodes=3))):
            envc = BatchedTaskEnv(B, A, T, device=DEV, **kw).load_instances(**inst)
            r, g = timed_run(envc, m, check_every=4, buckets=BUCKETS)
            rows[key] = dict(precision=k, **r, bucket_steps={str(a): b for a, b in g.bucket_steps.items()})
            envc.close()
        envs = BatchedTaskEnv(B, A, T, device=DEV, auto_reset=True).load_instances(**inst)
        g = GraphedRollout(envs, sampler(m), check_every=8)
        g.capture(seeds)
        envs.reset(seeds)
        for _ in range(40):
            g.graph.replay()
        d0 = int(envs.status()["decisions"].sum())
        sync(); t0 = time.perf_counter()
        for _ in range(400):
            g.graph.replay()
        sync(); wall = time.perf_counter() - t0
        dec = int(envs.status()["decisions"].sum()) - d0
        rows[f"tuned_{k}_steady_state_auto_reset"] = dict(precision=k, steps_per_s_end_to_end=dec / wall, batched_steps=400,
                                                          ms_per_batched_step=wall / 400 * 1e3, active_fraction=dec / (400 * B),
                                                          episodes=int(envs.episodes().sum()))
        envs.close()
        tunable.tuning_enable(True)
    tunable.tuning_enable(False)
    tunable.enable(False)
    fp32 = {k: v["steps_per_s_end_to_end"] for k, v in rows.items() if v["precision"] == "fp32"}
    best = max(fp32, key=fp32.get)
    return dict(config=3, workload=f"{B} envs {A}A/{T}T, attention policy (2.1M params, stock PyTorch) sampled + HIP env step",
                headline=dict(precision="fp32 (the reference's policy arithmetic)", row=best, steps_per_s=fp32[best],
                              one_episode_per_env=fp32.get("tuned_fp32"), three_episodes_per_env=fp32.get("tuned_fp32_3ep_auto_reset_compacted")),
                low_precision_agreement=agreement, **rows)


def config6(B=4096):
    """Runtime-shape path (training draws (agents_num, tasks_num) afresh every round, driver.py:114-115)."""
    from dcmrta_amd.instances import generate_batch_ranges
    rows = {}

    def rollout_rate(env, episodes=3, reps=8):
        for _ in range(3):
            env.rollout_random(episodes)
        sync(); t0 = time.perf_counter()
        counts = [env.rollout_random(episodes) for _ in range(reps)]
        sync()
        dt = time.perf_counter() - t0
        return int(torch.stack(counts).sum()) / dt

    def step_us(env, seeds, n=80):
        obs = env.reset(seeds)
        act = torch.zeros(env.B, dtype=torch.int32, device=DEV)
        ev = []
        for _ in range(n):
            act.copy_(torch.multinomial((~obs.mask).float(), 1).squeeze(1))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); obs = env.step(act); e1.record()
            ev.append((e0, e1))
        sync()
        return sorted(a.elapsed_time(b) for a, b in ev)[n // 2] * 1e3
    warm = BatchedTaskEnv(B, 20, 50, device=DEV).load_instances(**generate_batch(B, 20, 50, 0))   # clocks up before the first row
    warm.reset(env_seeds(0, 0, B), observe=False)
    for _ in range(30):
        warm.rollout_random(3)
    sync()
    warm.close()
    for name, (A, T) in (("20A50T exact <20,50>", (20, 50)), ("20A49T <20,50,runtime sizes>", (20, 49)),
                         ("15A35T <20,50,runtime sizes>", (15, 35)), ("21A51T <64,64,runtime sizes>", (21, 51)), ("70A130T mid-size <2,3 chunks>", (70, 130))):
        env = BatchedTaskEnv(B, A, T, device=DEV).load_instances(**generate_batch(B, A, T, 0))
        seeds = env_seeds(0, 0, B)
        env.reset(seeds, observe=False)
        rows[name] = dict(rollout_steps_per_s=rollout_rate(env), k_step_us_per_batched_step=step_us(env, seeds))
        env.close()
    rag = generate_batch_ranges(range(B), (10, 20), (20, 50))
    env = BatchedTaskEnv(B, 20, 50, device=DEV).load_instances(**rag)
    seeds = env_seeds(0, 0, B)
    env.reset(seeds, observe=False)
    rows["ragged (10-20)x(20-50) <20,50,runtime sizes>"] = dict(rollout_steps_per_s=rollout_rate(env),
                                                                k_step_us_per_batched_step=step_us(env, seeds))
    return dict(config=6, workload=f"{B} envs, runtime-shape instantiations vs the exact one (one stream, 3 episodes per launch)", **rows)


def config4(B=8192, A=50, T=200, reps=3):
    inst = generate_batch(B, A, T, 0)
    env = BatchedTaskEnv(B, A, T, device=DEV).load_instances(**inst)
    env.reset(env_seeds(0, 0, B), observe=False)
    env.rollout_random(1)
    sync(); t0 = time.perf_counter(); n = 0
    for _ in range(reps):
        n += int(env.rollout_random(1).sum())
    sync(); dt = time.perf_counter() - t0
    return dict(config=4, workload=f"{B} envs/GPU {A}A/{T}T random-policy rollout (shard of 65536 over 8 GPUs; the whole config: "
                                   f"bench.py --config 4 [--gpus N])",
                steps_per_s=n / dt, decisions_per_episode=n / reps / B)


def config5(B=1024, A=100, T=500):
    from dcmrta_amd.instances import synthetic_routes
    inst = generate_batch(B, A, T, 0)
    out = {}
    for reactive in (False, True):
        env = BatchedTaskEnv(B, A, T, device=DEV).load_instances(**inst)
        env.load_routes([synthetic_routes(inst["req"][b], A, max_task=100 if reactive else None) for b in range(B)], member_cap=5)
        env.execute_routes(reactive)
        sync(); t0 = time.perf_counter()
        r = env.execute_routes(reactive)
        sync(); dt = time.perf_counter() - t0
        out["reactive" if reactive else "static"] = dict(agent_steps_per_s=int(r["steps"].sum()) / dt,
                                                          mean_makespan=float(r["summary"][:, 3].mean()),
                                                          success_rate=float(r["summary"][:, 2].mean()))
    return dict(config=5, workload=f"{B} envs {A}A/{T}T route replay (execute_by_route), synthetic routes", **out,
                reference_python="~400 agent-steps/s at 20A/50T (BASELINE.md)")


def config7(A=20, T=50):
    """Runner level (what driver.py consumes): BatchedRunner.job = one sampled + one greedy episode per env + the experience
    buffers (runner.py:58-71, worker.py:41-112), decisions recorded per second of job wall time."""
    from dcmrta_amd.runner import BatchedRunner
    rows = {}
    for B, prec, twin, tune in ((256, "fp32", False, False), (256, "fp32", True, False), (4096, "fp32", False, False), (4096, "fp32", True, False),
                                (4096, "fp16", False, False), (4096, "fp32", False, True), (4096, "fp16", False, True)):
        torch.manual_seed(0)
        r = BatchedRunner(n_envs=B, device=DEV, rollout_precision=prec, twin_rollout=twin, tune_gemms=tune,
                          gemm_tuning_file=os.path.join(os.environ.get("TMPDIR", "/tmp"), "dcmrta_runner_tunableop.csv"))
        w = {k: v.clone() for k, v in r.get_weights().items()}
        r.job(w, w, 0, A, T)                                   # captures the graphs
        sync(); t0 = time.perf_counter()
        res, metrics, info = r.job(w, w, 1, A, T)
        sync(); wall = time.perf_counter() - t0
        n = res[0].shape[0]
        t0 = time.perf_counter()
        lists = [list(x.unbind(0)) if isinstance(x, torch.Tensor) else x for x in res]
        t_lists = time.perf_counter() - t0
        rows[f"B{B}_{prec}" + ("_twin_batch" if twin else "") + ("_tuned_gemms" if tune else "")] = dict(recorded_decisions=n, job_seconds=wall, recorded_decisions_per_s=n / wall,
                                    sim_decisions_per_s=(n + int(r.last["greedy_rec"]["active"].sum()) if r.last["greedy_rec"] else 2 * n) / wall,
                                    unbind_to_lists_seconds=t_lists, makespan=metrics["makespan"])
        r.close()
    import torch.cuda.tunable as tunable
    tunable.enable(False)
    return dict(config=7, workload=f"BatchedRunner.job at {A}A/{T}T (sampled episode + greedy twin + 9-slot experience per env)", **rows,
                reference_python="~80 recorded decisions/s per Ray worker process (BASELINE.md)")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="all")
    a = ap.parse_args()
    table = {"1": config1, "2l": config2_lockstep, "2g": config2_graph, "3": config3, "4": config4, "5": config5, "6": config6, "7": config7}
    for k in (table if a.config == "all" else [a.config]):
        print(json.dumps(table[k]()), flush=True)
