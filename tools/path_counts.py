#!/usr/bin/env python3
"""Dynamic path counts of the register-resident rollout kernel (developer tool, GPU box; behind profiles/r06_budget.md).

    python tools/path_counts.py [B A T episodes]          (default 4096 20 50 3 = BASELINE configs[1])

Builds tools/_variants/lib_cnt.so with -DDCM_COUNT_PATHS (CNT(i) marks in rollout_fast.hpp: one atomic per mark, lane 0) unless it
is already there, plays one pass and prints how often each path of the decision loop ran, per decision."""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SRC = os.path.join(ROOT, "dcmrta_amd", "csrc")
SO = os.path.join(ROOT, "tools", "_variants", "lib_cnt.so")
NAMES = {0: "decisions", 1: "follower draws", 2: "depot actions", 3: "re-join walks (Q4)", 4: "quiet joins (task_update skipped)",
         5: "task_update after a join", 6: "task_update calls after a join that remove members", 17: "task_update calls at a new event that remove members", 7: "dropping tasks visited", 8: "group exhausted",
         9: "next group of the same event", 10: "next_event (fast)", 12: "events with several deciders", 13: "events with several groups",
         14: "group-split iterations", 15: "general advance() calls", 16: "task_update calls with every task feasible"}


def main():
    B, A, T, EP = (int(x) for x in (sys.argv[1:5] + ["4096", "20", "50", "3"][len(sys.argv) - 1:]))
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    if not os.path.exists(SO):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-mllvm", "-phi-elim-split-all-critical-edges=1",
                               "-fPIC", "-shared", "-DDCM_COUNT_PATHS", os.path.join(SRC, "dcmrta_env.hip"),
                               os.path.join(SRC, "dcmrta_replay.hip"), "-o", SO])
    os.environ["DCMRTA_HIP_LIB"] = SO
    import torch
    from dcmrta_amd import _lib
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.choice import env_seeds
    from dcmrta_amd.instances import generate_batch
    lib = _lib.load()
    lib.dcm_debug_path_counts.restype = C.c_int
    lib.dcm_debug_path_counts.argtypes = [C.c_void_p]
    buf = (C.c_ulonglong * 32)()
    env = BatchedTaskEnv(B, A, T, device="cuda:0")
    env.load_instances(**generate_batch(B, A, T, base_seed=0))
    env.reset(env_seeds(0, 0, B), observe=False)
    _lib.check(lib.dcm_debug_path_counts(buf))                       # clear
    steps = env.rollout_random(episodes=EP)
    torch.cuda.synchronize()
    _lib.check(lib.dcm_debug_path_counts(buf))
    n = int(steps.sum().item())
    out = {"shape": [B, A, T, EP], "decisions": n, "per_decision": {NAMES.get(i, str(i)): round(buf[i] / n, 4) for i in range(32) if buf[i]}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
