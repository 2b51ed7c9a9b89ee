#!/usr/bin/env python3
"""Randomised parity sweep of the route-replay kernel against the oracle (developer tool, GPU box).

    python tools/sweep_replay.py [shapes] [seed] [auto|lds|hbm] [member_cap]

placement auto + member_cap <= 8 runs the register-resident kernel (replay_fast.hpp) where the shape allows it; lds / hbm ask for the
general kernel and say where it keeps its replay scratch (dcm_set_replay_placement).  Envs that overflow member_cap are skipped."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from dcmrta_amd.batched_env import BatchedTaskEnv  # noqa: E402
from dcmrta_amd.instances import generate_batch  # noqa: E402

n_shapes = int(sys.argv[1]) if len(sys.argv) > 1 else 100
B = 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 77)
placement = sys.argv[3] if len(sys.argv) > 3 else "auto"
member_cap = int(sys.argv[4]) if len(sys.argv) > 4 else 12
KEYS = ("finished", "time_start", "time_finish", "task_wait", "n_members", "travel_dist", "returned", "agent_wait")
bad = checked = trunc = terr = skipped = 0
t0 = time.time()
for it in range(n_shapes):
    A = int(rng.choice([2, 3, 5, 8, 13, 20, 33, 40, 64, 70]))
    T = int(rng.choice([3, 7, 20, 37, 50, 64, 65, 90]))
    reactive = bool(rng.integers(0, 2))
    inst = generate_batch(B, A, T, base_seed=int(rng.integers(0, 1 << 30)))
    if it % 2:
        inst["dur"] = rng.random((B, T)) * 5.0
    routes = []
    for b in range(B):
        r = [[] for _ in range(A)]
        for t in range(T):
            k = min(A, int(inst["req"][b][t]) + int(rng.integers(0, 3)) - int(rng.integers(0, 2)))   # sometimes too few, sometimes extra
            for a in rng.choice(A, size=max(k, 0), replace=False):
                r[int(a)].append(t + 1)
        rr = []
        for a in range(A):
            if rng.random() < 0.05:
                rr.append(None)                       # pre_set_route stays None
                continue
            x = r[a]
            if rng.random() < 0.5:
                x = sorted(x)
            else:
                rng.shuffle(x)
            rr.append([int(v) for v in x] + ([0] if rng.random() < 0.8 else []))
        routes.append(rr)
    env = BatchedTaskEnv(B, A, T).load_instances(**inst)
    env.load_routes(routes, member_cap=member_cap)
    env.set_replay_placement(placement)
    out = env.execute_routes(reactive=reactive)
    flags = out["flags"].cpu().numpy()
    for b in range(B):
        o = oracle.OracleEnv(A, T).load(inst["depot"][b], inst["task_xy"][b], inst["req"][b], inst["dur"][b])
        for a, r in enumerate(routes[b]):
            if r is not None:
                o.pre_set_route(r, a)
        checked += 1
        if flags[b] & 16:
            # the kernel stops where the member slots run out: nothing to compare -- but the oracle must agree that some task
            # really lists more than member_cap members (a kernel that sets the flag wrongly would otherwise just shrink the sweep)
            skipped += 1
            try:
                ref = o.execute_by_route(reactive)
                if ref["max_members_seen"] <= member_cap:
                    bad += 1; print("MISMATCH overflow flag without an overflow", A, T, reactive, b, ref["max_members_seen"], member_cap)
            except TypeError:
                pass                                                # (the reference raises later in the episode: undecidable here)
            continue
        try:
            ref = o.execute_by_route(reactive)
        except TypeError:
            terr += 1
            if not (flags[b] & 64):
                bad += 1; print("MISMATCH type-error not flagged", A, T, reactive, b)
            continue
        if flags[b] & 64:
            bad += 1; print("MISMATCH spurious type error", A, T, reactive, b); continue
        if bool(flags[b] & 4) != bool(ref["truncated"]):
            bad += 1; print("MISMATCH truncated", A, T, reactive, b, flags[b], ref["truncated"]); continue
        trunc += int(ref["truncated"])
        for k in KEYS:
            got = out[k][b].cpu().numpy()
            if not np.array_equal(got.astype(np.asarray(ref[k]).dtype), ref[k]):
                bad += 1; print("MISMATCH", k, A, T, reactive, b); break
        else:
            sm = out["summary"][b].cpu().numpy()
            if not np.array_equal(sm[2:8], ref["metrics"]):
                bad += 1; print("MISMATCH metrics", A, T, reactive, b, sm[2:8], ref["metrics"])
                arrs = [o.route(a)[1] for a in range(A)]
                print("   flags", flags[b], "oracle truncated", ref["truncated"], "max last arrival", max([x[-1] for x in arrs if len(x)] + [0]),
                      "max any arrival", max([x.max() for x in arrs if len(x)] + [0]), "returned", int(ref["returned"].sum()), "/", A,
                      "finished", int(ref["finished"].sum()), "/", T, "routes", [len(r) if r is not None else None for r in routes[b]][:8])
    env.close()
print(f"replay sweep: {n_shapes} shapes, {checked} episodes, {bad} mismatches, {trunc} truncated, {terr} type-errors, "
      f"{skipped} skipped (member_cap overflow, confirmed by the oracle), {time.time()-t0:.0f} s")
