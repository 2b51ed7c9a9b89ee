#!/usr/bin/env python3
"""Golden vectors for the PARAMETRISED dynamic-arrival schedule (run in the build container only).

    python tests/golden/make_golden_schedule.py   ->  tests/golden/replay_schedule.json

The reference hard-codes the schedule of execute_by_route's reactive mode: 20 tasks visible at the start, +20 every 10 time
units, never more than 100 (env/task_env.py:567), and the matching depot re-arm time (next - 1) // 20 * 10 (:221).  With
those constants tasks 101..500 of a 100A/500T instance never become visible.  The build keeps the
four constants as parameters (dcm_set_visibility, defaults = the reference).  To pin the generalisation against the
reference itself, this script loads the reference module's text AT GENERATION TIME, substitutes exactly those literals in
those two lines, executes the result in memory and replays routes through it -- i.e. "the reference with four constants
changed".  Nothing of the reference is written anywhere: the fixture holds sizes, seeds, routes and result numbers only.
"""
import contextlib
import copy
import io
import json
import os
import signal
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

SRC = os.path.join(mg.REF, "env", "task_env.py")
LINE_VISIBLE = "self.current_time//10 * 20 + 20, 20, 100"      # env/task_env.py:567
LINE_REARM = "(next_action - 1)//20 * 10"                       # env/task_env.py:221


def patched_taskenv(initial, batch, period, cap):
    """TaskEnv class of the reference with the four schedule literals replaced (in memory only)."""
    text = open(SRC).read()
    assert text.count(LINE_VISIBLE) == 1 and text.count(LINE_REARM) == 1, "reference source changed: re-derive the two lines"
    text = text.replace(LINE_VISIBLE, f"self.current_time//{period} * {batch} + {initial}, {initial}, {cap}")
    text = text.replace(LINE_REARM, f"(next_action - 1)//{batch} * {period}")
    mod = types.ModuleType("task_env_patched")
    mod.__file__ = SRC
    exec(compile(text, SRC, "exec"), mod.__dict__)
    return mod.TaskEnv


def nearest_partition_routes(req, A, max_task=None):
    """Same synthetic routes as the config-5 benchmark (dcmrta_amd/instances.py::synthetic_routes)."""
    T = len(req)
    r = [[] for _ in range(A)]
    for t in range(T if max_task is None else min(T, max_task)):
        for j in range(int(req[t])):
            r[(7 * t + j) % A].append(t + 1)
    return [x + [0] for x in r]


class _Timeout(Exception):
    pass


def main():
    def on_alarm(sig, frm):
        raise _Timeout()
    signal.signal(signal.SIGALRM, on_alarm)
    rng = np.random.default_rng(2024)
    cases = []
    # (A, T, (initial, batch, period, cap), route style)
    specs = [
        (20, 50, (10, 10, 5, 50), "synthetic"), (20, 50, (5, 15, 7, 40), "synthetic"), (20, 50, (20, 20, 10, 100), "synthetic"),
        (30, 120, (20, 20, 10, 120), "synthetic"), (30, 120, (30, 30, 4, 120), "random"), (13, 37, (8, 4, 3, 37), "random"),
        (50, 200, (40, 40, 10, 200), "synthetic"), (50, 200, (20, 20, 2, 200), "random"), (10, 64, (16, 16, 16, 48), "random"),
        (100, 500, (100, 100, 10, 500), "synthetic"), (100, 500, (20, 20, 10, 100), "synthetic100"),
        (30, 120, (30, 30, 4, 120), "random_all"), (13, 37, (8, 4, 3, 37), "random_all"), (50, 200, (20, 20, 2, 200), "random_all"),
        (20, 50, (7, 13, 3, 50), "random_all"),
    ]
    out_path = os.path.join(HERE, "replay_schedule.json")
    if os.path.exists(out_path) and "--append" in sys.argv:      # keep the cases already generated, add the new specs
        cases = json.load(open(out_path))
    for ci, (A, T, sched, style) in enumerate(specs):
        if ci < len(cases):
            continue
        TaskEnv = patched_taskenv(*sched)
        env = TaskEnv((A, A), (T, T), 1, 5, seed=500 + ci)
        ia = mg.instance_arrays(env)
        if style == "synthetic":
            routes = nearest_partition_routes(ia["req"], A)
        elif style == "synthetic100":
            routes = nearest_partition_routes(ia["req"], A, max_task=100)
        else:
            import make_golden_extra as mge
            routes = mge.random_routes(rng, A, T, ia["req"])
            if style == "random_all":        # every agent has a route (no pre_set_route None -> no TypeError at :220)
                routes = [r if r is not None else [0] for r in routes]
        env.reactive_planning = True
        for a, r in enumerate(routes):
            if r is not None:
                env.pre_set_route(copy.copy(r), a)
        case = dict(A=A, T=T, inst_seed=500 + ci, schedule=list(sched), routes=routes, req_sum=int(ia["req"].sum()), style=style)
        signal.alarm(300)
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                env.execute_by_route("./", "x", False)
            env.get_episode_reward(100)
            fa = mg.final_arrays(env)
            case["result"] = {k: np.asarray(fa[k]).tolist() for k in
                              ("metrics", "finished", "time_start", "time_finish", "task_wait", "agent_wait", "travel_dist",
                               "returned", "n_members", "route_len")}
            case["status"] = "ok"
        except TypeError:
            case["status"] = "type_error"
        except _Timeout:
            case["status"] = "no_termination"
        finally:
            signal.alarm(0)
        print("schedule", ci, A, T, sched, style, case["status"], case.get("result", {}).get("metrics", [None, None])[:2], flush=True)
        cases.append(case)
    with open(out_path, "w") as f:
        json.dump(cases, f)


if __name__ == "__main__":
    main()
