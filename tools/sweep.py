#!/usr/bin/env python3
"""Randomised parity sweep on the GPU box: many (A, T, seed) shapes, persistent rollout kernel (one launch and budgeted launches
kernel) and lockstep API against the oracle.  Developer tool; prints the number of envs checked and any mismatch."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402
import oracle  # noqa: E402
from dcmrta_amd.batched_env import BatchedTaskEnv  # noqa: E402
from dcmrta_amd.choice import env_seeds  # noqa: E402
from dcmrta_amd.instances import generate_batch  # noqa: E402

n_shapes = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 2026)
TASK_KEYS = ("finished", "feasible", "time_start", "time_finish", "task_wait", "n_members", "n_abandoned")
AGENT_KEYS = ("travel_dist", "returned", "agent_wait")
bad, checked, t0 = 0, 0, time.time()
wait_order = 0
is_cnt = 0
masked_cnt = 0
auto_cnt = 0
for it in range(n_shapes):
    A = int(rng.choice([1, 2, 3, 5, 8, 13, 20, 31, 32, 33, 50, 63, 64, 65, 100, 128]))
    T = int(rng.choice([1, 2, 7, 20, 37, 50, 63, 64, 65, 100, 128, 129, 200, 300]))
    mwt = float(rng.choice([10.0, 10.0, 3.0, 25.0]))
    max_time = float(rng.choice([100.0, 100.0, 30.0, 250.0]))
    base = int(rng.integers(0, 1 << 30))
    inst = generate_batch(B, A, T, base_seed=base)
    if it % 3 == 0:   # non-constant durations like the shipped test set (U(0,5))
        inst["dur"] = rng.random((B, T)) * 5.0
    elif it % 7 == 1:   # zero / long durations
        inst["dur"] = np.full((B, T), float(rng.choice([0.0, 20.0])))
    if it % 5 == 4:     # smaller coalitions (max_coalition_size 1..3) or all-5 requirements
        inst["req"] = rng.integers(1, int(rng.integers(1, 4)) + 1, (B, T)).astype(np.int32) if it % 2 else np.full((B, T), 5, np.int32)
    if it % 11 == 5:    # coincident task locations / a task on the depot
        inst["task_xy"][:, 1::2] = inst["task_xy"][:, 0:1]
        inst["task_xy"][:, -1] = inst["depot"]
    seeds = env_seeds(base ^ 0x5A5A, 0, B)
    ragged = (it % 5 == 2)   # every env of the batch gets its own (A_e, T_e) <= (A, T): dcm_load_instances_ragged
    nA = rng.integers(1, A + 1, B).astype(np.int32) if ragged else np.full(B, A, np.int32)
    nT = rng.integers(1, T + 1, B).astype(np.int32) if ragged else np.full(B, T, np.int32)
    if ragged:
        inst["n_agents"], inst["n_tasks"] = nA, nT
    refs = []
    for b in range(B):
        a, t = int(nA[b]), int(nT[b])
        o = oracle.OracleEnv(a, t, max_waiting_time=mwt, max_time=max_time).load(inst["depot"][b], inst["task_xy"][b, :t], inst["req"][b, :t], inst["dur"][b, :t])
        refs.append(o.rollout(int(seeds[b]), 0, oracle.POLICY_RANDOM, cap_steps=100000, record=False))
    for mode in ("rollout", "budget", "lockstep"):
        env = BatchedTaskEnv(B, A, T, max_waiting_time=mwt, max_time=max_time).load_instances(**inst)
        if mode == "lockstep":
            if it % 4 and not (ragged and it % 2):
                continue
            # every per-decision output (leader, event time, mask, both observation tensors) against the oracle's trace
            got = H.run_lockstep(env, seeds, lambda b, i, m, l: H.host_random_action(m, int(seeds[b]), i))
            steps = np.array([g["n_steps"] for g in got], np.int64)
            for b in range(B):
                a, t = int(nA[b]), int(nT[b])
                o = oracle.OracleEnv(a, t, max_waiting_time=mwt, max_time=max_time).load(inst["depot"][b], inst["task_xy"][b, :t], inst["req"][b, :t], inst["dur"][b, :t])
                ref = o.rollout(int(seeds[b]), 0, oracle.POLICY_RANDOM, cap_steps=100000, record=True)
                g = got[b]
                ok = (g["n_steps"] == ref["n_steps"] and np.array_equal(g["leader"], ref["leader"]) and np.array_equal(g["now"], ref["now"])
                      and np.array_equal(g["mask"][:, :t + 1], ref["mask"]) and np.array_equal(g["agents_obs"][:, :a], ref["agents_obs"])
                      and np.array_equal(g["tasks_obs"][:, :t + 1], ref["tasks_obs"])
                      and (g["mask"][:, t + 1:] == 1).all() and (g["agents_obs"][:, a:] == -1).all() and (g["tasks_obs"][:, t + 1:] == -1).all())
                if not ok:
                    bad += 1
                    print("MISMATCH per-step outputs", A, T, mwt, base, b, "ragged", ragged, flush=True)
        elif mode == "budget":
            # the same episode in budgeted launches (random per-env decision budgets): stops anywhere, carries on identically
            env.reset(seeds, observe=False)
            steps = np.zeros(B, np.int64)
            nref = np.array([r["n_steps"] for r in refs], np.int64)
            for _ in range(3):   # (budgets stay inside the episode: a finished env would start its next episode)
                bud = np.clip(np.minimum(rng.integers(0, 60, B), nref - 1 - steps), 0, None).astype(np.int64)
                steps += env.rollout_random(1, max_decisions=bud).cpu().numpy()
            steps += env.rollout_random(1).cpu().numpy()
        else:
            env.reset(seeds, observe=False)
            steps = env.rollout_random(1).cpu().numpy()
        fin = H.gpu_final(env)
        for b in range(B):
            try:
                assert steps[b] == refs[b]["n_steps"], ("steps", steps[b], refs[b]["n_steps"])
                f = dict(fin[b])
                for k in TASK_KEYS:
                    f[k] = f[k][:nT[b]]
                for k in AGENT_KEYS:
                    f[k] = f[k][:nA[b]]
                wait_order += 1 if f["flags"] & 128 else 0   # DCM_FLAG_WAIT_ORDER (saturated abandonment counter): not expected
                H.assert_final_matches(f, refs[b], f"{mode} {A}A{T}T mwt={mwt} base={base} env{b} ragged={ragged}")
            except AssertionError as ex:
                bad += 1
                print("MISMATCH", mode, A, T, mwt, base, b, str(ex)[:200], flush=True)
            checked += 1
        env.close()
    # a collection loop: DCM_PARAM_AUTO_RESET lockstep with a device-side first-valid policy and no host read until the end (the
    # register-resident step parks final records and restarts from the reset image; k_terminal_flush computes the summaries):
    # every episode's return (return log) and the last episode's metrics against the oracle's consecutive episodes
    if it % 3 == 1 and A <= 64 and T <= 64:
        nB, n_steps, cap = min(B, 24), 360, 512
        sub = {k: v[:nB] for k, v in inst.items()}
        env = BatchedTaskEnv(nB, A, T, max_waiting_time=mwt, max_time=max_time, auto_reset=True).load_instances(**sub)
        ring = env.enable_return_log(cap)
        obs = env.reset(seeds[:nB])
        for _ in range(n_steps):
            obs = env.step(torch.argmax((~obs.mask).to(torch.int32), dim=1).to(torch.int32))
        eps = env.episodes().cpu().numpy()
        sm = env.summary().cpu().numpy()
        rl = ring.cpu().numpy()
        dec = env.status()["decisions"].cpu().numpy()
        for b in range(nB):
            a, t = int(nA[b]), int(nT[b])
            o = oracle.OracleEnv(a, t, max_waiting_time=mwt, max_time=max_time).load(inst["depot"][b], inst["task_xy"][b, :t], inst["req"][b, :t], inst["dur"][b, :t])
            d0, ok, last = 0, int(eps[b]) <= cap, None
            for k in range(min(int(eps[b]), cap)):
                last = o.rollout(int(seeds[b]), d0, oracle.POLICY_FIRST, cap_steps=100000, record=False)
                o.clear_decisions()
                d0 += last["n_steps"]
                ok = ok and rl[b, k] == last["reward"]
            if last is not None:
                ok = ok and sm[b, 0] == last["reward"] and all(sm[b, 2 + i] == last["metrics"][i] or (np.isnan(sm[b, 2 + i]) and np.isnan(last["metrics"][i])) for i in range(6))
            ok = ok and d0 <= n_steps and dec[b] == n_steps
            checked += max(int(eps[b]), 1)
            auto_cnt += int(eps[b])
            if not ok:
                bad += 1
                print("MISMATCH auto-reset lockstep", A, T, mwt, max_time, base, b, "ragged", ragged, int(eps[b]), flush=True)
        env.close()
    # individual selection (Worker.run_test_IS, worker.py:159-198): the device offers the lowest pending id and moves it alone;
    # the reference loop is restated on the oracle's step-wise surface with the same keyed valid-action choice
    if it % 9 == 4 and A <= 40 and T <= 70 and not ragged:
        nB = min(B, 4)
        env = BatchedTaskEnv(nB, A, T, max_waiting_time=mwt, max_time=max_time, individual_selection=True)
        env.load_instances(**{k: v[:nB] for k, v in inst.items()})
        got = H.run_lockstep(env, seeds[:nB], lambda b, i, m, l: H.host_random_action(m, int(seeds[b]), i))
        fin = H.gpu_final(env)
        for b in range(nB):
            o = oracle.OracleEnv(A, T, max_waiting_time=mwt, max_time=max_time).load(inst["depot"][b], inst["task_xy"][b], inst["req"][b], inst["dur"][b])
            n, finished, guard, ok = 0, False, 0, True
            while not finished and o.now < max_time and guard < 100000:
                ids, t = o.next_decision()
                o.now = t
                o.task_update(); o.agent_update()
                for a in ids:
                    g = got[b]
                    if n >= g["n_steps"] or int(g["leader"][n]) != int(a) or not np.array_equal(g["mask"][n], o.mask()):
                        ok = False
                        break
                    o.agent_step(int(a), int(g["action"][n]))
                    o.task_update(); o.agent_update()
                    n += 1
                if not ok:
                    break
                finished = o.check_finished()
                guard += 1 if len(ids) else 20000          # zero-decider events: the shared guard ends the episode
            checked += 1
            is_cnt += 1
            if ok:
                oracle.lib().orc_finish_episode(o._h)
                ref = o.final()
                ok = n == got[b]["n_steps"] and (fin[b]["flags"] & 4 or (float(fin[b]["reward"]) == ref["reward"] and
                                                                             np.array_equal(fin[b]["travel_dist"], ref["travel_dist"]) and np.array_equal(fin[b]["time_start"], ref["time_start"])))
            if not ok:
                bad += 1
                print("MISMATCH individual-selection", A, T, mwt, base, b, flush=True)
        env.close()
    # a policy that ignores the mask (TaskEnv.step simulates masked actions, env/task_env.py:326-342): lockstep API with the host
    # mirror of ORC_POLICY_ANY; an env may only freeze (DCM_FLAG_OVERFLOW) when the oracle's longest member list exceeds 5
    if it % 6 == 1 and A <= 64 and T <= 130 and not ragged:
        nB = min(B, 12)
        env = BatchedTaskEnv(nB, A, T, max_waiting_time=mwt, max_time=max_time).load_instances(**{k: v[:nB] for k, v in inst.items()})

        def anymask(b, i, m, l):
            from dcmrta_amd.choice import below, draw
            r = draw(int(seeds[b]), i, 1)
            if r % 16 == 1:
                return 0
            if r % 4 == 0:
                return 1 + (r >> 4) % T
            valid = np.flatnonzero(m == 0)
            return int(valid[below(r, len(valid))])
        try:
            got = H.run_lockstep(env, seeds[:nB], anymask, max_iters=20000)
        except RuntimeError:
            got = None                                     # (time can step backwards: an episode may not end within the cap)
        if got is not None:
            fin = H.gpu_final(env)
            for b in range(nB):
                o = oracle.OracleEnv(A, T, max_waiting_time=mwt, max_time=max_time).load(inst["depot"][b], inst["task_xy"][b], inst["req"][b], inst["dur"][b])
                try:
                    ref = o.rollout(int(seeds[b]), 0, oracle.POLICY_ANY, cap_steps=20000, record=True)
                except RuntimeError:
                    continue
                g, n = got[b], got[b]["n_steps"]
                over = bool(fin[b]["flags"] & 16)
                ok = over == (ref["max_members_seen"] > 5) and all(np.array_equal(g[k], ref[k][:n]) for k in ("leader", "now", "mask", "agents_obs", "tasks_obs"))
                if ok and not over:
                    try:
                        assert n == ref["n_steps"]
                        H.assert_final_matches(fin[b], ref, "anymask")
                    except AssertionError:
                        ok = False
                checked += 1
                masked_cnt += 1
                if not ok:
                    bad += 1
                    print("MISMATCH mask-ignoring policy", A, T, mwt, base, b, flush=True)
        env.close()
print(f"sweep: {n_shapes} shapes, {checked} env-episodes checked, {bad} mismatches, {wait_order} with the wait-order flag, {is_cnt} in individual-selection mode, {masked_cnt} under a mask-ignoring policy, {auto_cnt} episodes of auto-resetting lockstep loops, {time.time() - t0:.0f} s")
