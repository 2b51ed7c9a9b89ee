#!/usr/bin/env python3
"""Fresh-case fuzz of the C oracle against the IMPORTED reference (build container only; never collected by pytest, never run on
the GPU box: it needs /root/reference).

    python tests/golden/fuzz_oracle.py --cases 240 --seed 606         ->  one summary line (and one line per mismatch)

The goldens under tests/golden/ are a fixed set; this re-pins oracle/dcmrta_oracle.c on cases no fixture contains.  Per case:
random A <= 70, T <= 110, max coalition size 1..5, max_waiting_time in {10, 3, 25}, then one of
  * an RL-mode episode (worker.py:45-87 through make_golden.rollout, the harness behind every trace fixture) under the
    uniform-random / first-valid / nearest-valid policy or the mask-IGNORING one of make_golden_masked.py, compared on step count,
    reward and the sha256 digest of leader / action / followers / time / mask / both observation tensors / metrics / terminal arrays;
  * a route replay (env/task_env.py:562-599, make_golden_extra.random_routes) with reactive_planning False or True, compared on
    every terminal array, on TypeError where the reference raises it (:220) and on the zero-decider guard where it never ends.
Nothing is written; the run's summary line goes into profiles/README.md by hand.
"""
import argparse
import contextlib
import copy
import io
import os
import signal
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import make_golden as mg  # noqa: E402   (imports the reference read-only)
import make_golden_extra as mge  # noqa: E402
import make_golden_masked as mgm  # noqa: E402
import helpers as H  # noqa: E402
import oracle  # noqa: E402

ORC_POLICY = {"random": oracle.POLICY_RANDOM, "first": oracle.POLICY_FIRST, "nearest": oracle.POLICY_NEAREST, "anymask": oracle.POLICY_ANY}
REPLAY_KEYS = ("metrics", "finished", "time_start", "time_finish", "task_wait", "agent_wait", "travel_dist", "returned", "n_members",
               "route_len")


class _Timeout(Exception):
    pass


def _on_alarm(sig, frm):
    raise _Timeout()


def rl_case(rng, ci, counts):
    A, T = int(rng.integers(1, 71)), int(rng.integers(1, 111))
    coal = int(rng.integers(1, 6))
    mwt = float(rng.choice([10.0, 10.0, 3.0, 25.0]))
    inst_seed = int(rng.integers(0, 1 << 31))
    pol = ("random", "first", "nearest", "anymask")[ci % 4]
    env = mg.TaskEnv((A, A), (T, T), 1, coal, seed=inst_seed)          # env/task_env.py:9-34
    env.max_waiting_time = mwt
    ia = mg.instance_arrays(env)
    seed_e = mg.env_seed(int(rng.integers(0, 1 << 40)), ci)
    if pol == "anymask":
        try:
            tr, st = mgm.rollout_anymask(env, seed_e)
        except mgm._TooLong:
            counts["skipped (anymask episode over the decision cap)"] += 1
            return None
        if int(tr["truncated"]):
            counts["skipped (reference never terminates)"] += 1
            return None
        cap = 4000
    else:
        tr = mg.rollout(env, seed_e, mg.POLICIES[pol])
        cap = 8192
    o = oracle.OracleEnv(A, T, max_waiting_time=mwt).load(ia["depot"], ia["task_xy"], ia["req"], ia["dur"])
    out = o.rollout(int(seed_e), 0, ORC_POLICY[pol], cap_steps=cap)
    ok = out["n_steps"] == int(tr["n_steps"]) and out["reward"] == float(tr["reward"]) and H.digest(out) == mg.digest(tr)
    counts[f"rl:{pol}"] += 1
    return ok, f"RL A={A} T={T} coal={coal} mwt={mwt} inst_seed={inst_seed} seed_e={seed_e} policy={pol} steps {out['n_steps']} vs {int(tr['n_steps'])}"


def replay_case(rng, ci, counts):
    A, T = int(rng.integers(1, 71)), int(rng.integers(1, 111))
    coal = int(rng.integers(1, 6))
    inst_seed = int(rng.integers(0, 1 << 31))
    reactive = bool(ci & 1)
    env = mg.TaskEnv((A, A), (T, T), 1, coal, seed=inst_seed)
    ia = mg.instance_arrays(env)
    routes = mge.random_routes(rng, A, T, ia["req"])
    env.reactive_planning = reactive
    for a, r in enumerate(routes):
        if r is not None:
            env.pre_set_route(copy.copy(r), a)                           # env/task_env.py:595-599
    signal.alarm(60)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            env.execute_by_route("./", "x", False)                       # :562-593
        env.get_episode_reward(100)
        fa = mg.final_arrays(env)
        status = "ok"
    except TypeError:
        status = "type_error"
    except _Timeout:
        status = "no_termination"
    finally:
        signal.alarm(0)
    o = oracle.OracleEnv(A, T).load(ia["depot"], ia["task_xy"], ia["req"], ia["dur"])
    for a, r in enumerate(routes):
        if r is not None:
            o.pre_set_route(r, a)
    try:
        ref = o.execute_by_route(reactive)
        ostatus = "no_termination" if ref["truncated"] else "ok"
    except TypeError:
        ostatus = "type_error"
    ok = ostatus == status
    if ok and status == "ok":
        for k in REPLAY_KEYS:
            exp = np.asarray(fa[k])
            ok = ok and np.array_equal(np.asarray(ref[k]).astype(exp.dtype), exp, equal_nan=True)
    counts[f"replay:{'reactive' if reactive else 'static'}:{status}"] += 1
    return ok, f"REPLAY A={A} T={T} coal={coal} inst_seed={inst_seed} reactive={reactive} status {ostatus} vs {status}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=240)
    ap.add_argument("--seed", type=int, default=606)
    args = ap.parse_args()
    signal.signal(signal.SIGALRM, _on_alarm)
    oracle.build()
    rng = np.random.default_rng(args.seed)
    import collections
    counts = collections.Counter()
    bad = compared = 0
    t0 = time.time()
    for ci in range(args.cases):
        r = (replay_case if ci % 3 == 2 else rl_case)(rng, ci, counts)
        if r is None:
            continue
        compared += 1
        if not r[0]:
            bad += 1
            print("MISMATCH", r[1], flush=True)
    kinds = ", ".join(f"{k} {v}" for k, v in sorted(counts.items()))
    print(f"fuzz_oracle: seed {args.seed}, {args.cases} cases, {compared} compared, {bad} mismatches, {time.time() - t0:.0f} s [{kinds}]")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
