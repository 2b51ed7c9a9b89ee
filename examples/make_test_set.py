#!/usr/bin/env python3
"""TestSetGenerator.py of the reference (TestSetGenerator.py:1-116), without the env class: instance i of a test set is
TaskEnv(agents_range, tasks_range, traits_dim=1, max_coalition_size=5, seed=i) (:16-18), i.e. a seeded draw in the order of
env/task_env.py:57-114; the script writes the instances as one npz (the format of tests/golden/instances_20A50T.npz, read by
dcmrta_amd.instances.load_instances_npz / examples/rl_test.py) and, per instance, the four CTAS-D planner input files
(vehicle_param / task_param / planner_param / graph yaml, :51-116) so that the external MILP baseline can be run on them.

    python examples/make_test_set.py --out testSet_20A_50T_CONDET [--num 50] [--agents 20] [--tasks 50] [--no-yaml]

With the defaults the depot and task coordinates are those of the reference's shipped testSet_20A_50T_CONDET (same seeds, same
first draws; checked by tests/test_host.py).  The shipped pickles themselves were written by an older generator (random task
durations, another requirement draw), so their requirements / durations are NOT what the current TaskEnv(seed=i) produces: use
tests/golden/instances_20A50T.npz (read out of those pickles) to evaluate on the published test set itself.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dcmrta_amd.ctasd_io import export_ctasd_yaml  # noqa: E402
from dcmrta_amd.instances import generate_instance_ranges  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True, help="folder of the test set (TestSetGenerator.py:10 `test_set`)")
    ap.add_argument("--num", type=int, default=50)          # test_instances_num :11
    ap.add_argument("--agents", type=int, default=20)       # agents_range = (A, A) :12
    ap.add_argument("--tasks", type=int, default=50)        # tasks_range = (T, T) :13
    ap.add_argument("--solver-time", type=float, default=300.0)
    ap.add_argument("--no-yaml", action="store_true")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    folder = os.path.basename(os.path.normpath(a.out))
    insts = []
    for i in range(a.num):
        # tuple ranges, as the reference passes them: the sizes are drawn from the seeded stream first (env/task_env.py:58-65)
        A, inst = generate_instance_ranges((a.agents, a.agents), (a.tasks, a.tasks), i, max_coalition_size=5)
        insts.append(inst)
        if not a.no_yaml:
            export_ctasd_yaml(os.path.join(a.out, f"env_{i}"), inst["depot"], inst["task_xy"], inst["req"], inst["dur"], A,
                              folder=folder, index=i, solver_time=a.solver_time)
    path = os.path.join(a.out, f"instances_{a.agents}A{a.tasks}T.npz")
    np.savez_compressed(path, depot=np.stack([x["depot"] for x in insts]), task_xy=np.stack([x["task_xy"] for x in insts]),
                        req=np.stack([x["req"] for x in insts]), dur=np.stack([x["dur"] for x in insts]), A=np.int64(a.agents))
    print(f"{a.num} instances of {a.agents}A/{a.tasks}T -> {path}" + ("" if a.no_yaml else f" + {a.num} x 4 CTAS-D yaml files"))


if __name__ == "__main__":
    main()
