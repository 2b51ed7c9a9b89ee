#!/usr/bin/env python3
"""Shader clocks per phase of k_replay's event loop (developer tool; needs a -DDCM_REPLAY_PHASES build of the library):

    DCMRTA_HIP_LIB=<phase build> python tools/replay_phases.py [B A T] [a,b,c,d | static]

prints, per agent step, the clocks spent in: event prologue, full task_update, single-task task_update, agent_update,
agent_step, check_finished -- mean over envs."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dcmrta_amd.batched_env import BatchedTaskEnv  # noqa: E402
from dcmrta_amd.instances import generate_batch, synthetic_route_arrays  # noqa: E402

args = sys.argv[1:]
vis = (20, 20, 10, 100)
reactive = True
if args and ("," in args[-1] or args[-1] == "static"):
    v = args.pop()
    if v == "static":
        reactive = False
    else:
        vis = tuple(int(x) for x in v.split(","))
B, A, T = (int(x) for x in args[:3]) if len(args) >= 3 else (768, 100, 500)
inst = generate_batch(B, A, T, base_seed=0, first=0)
env = BatchedTaskEnv(B, A, T, device="cuda:0")
env.load_instances(**inst)
routes, route_len = synthetic_route_arrays(inst["req"], A, max_task=min(T, vis[3]) if reactive else None)
env.set_visibility(*vis)
env.load_route_arrays(routes, route_len, member_cap=5)
for _ in range(2):
    out = env.execute_routes(reactive, fields=("time_start",))
torch.cuda.synchronize()
ph = out["time_start"][:, :12].cpu().numpy()
steps = out["steps"].cpu().numpy().astype(np.float64)
names = {0: "prologue", 1: "task_update(full)", 2: "task_update(one)", 3: "agent_update(event)", 4: "agent_step", 5: "check_finished",
         8: "agent_update(inline)", 9: "agent_update(one)", 10: "agent_update(task/full)"}
tot = ph[:, list(names)].sum(1)
print(f"B={B} {A}A/{T}T visibility={vis if reactive else 'static'}: {steps.mean():.0f} agent steps, {ph[:, 6].mean():.0f} events and {ph[:, 11].mean():.0f} task/full agent updates after a step per env; "
      f"{tot.mean() / steps.mean():.0f} clocks per agent step")
for i, n in names.items():
    print(f"  {n:20s} {ph[:, i].mean() / steps.mean():8.0f} clocks/step   {ph[:, i].sum() / tot.sum():6.1%}")
