#!/usr/bin/env python3
"""k_replay launch time with the replay scratch in LDS vs HBM (developer tool, GPU box):  python tools/replay_placement_ab.py [B ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dcmrta_amd.batched_env import BatchedTaskEnv  # noqa: E402
from dcmrta_amd.instances import generate_batch, synthetic_route_arrays  # noqa: E402

A, T = 100, 500
for B in [int(x) for x in sys.argv[1:]] or [1024, 2048, 8192]:
    inst = generate_batch(B, A, T, base_seed=0, first=0)
    for reactive in (True, False):
        routes, route_len = synthetic_route_arrays(inst["req"], A, max_task=100 if reactive else None)
        row = {}
        for pl in ("lds", "hbm", "auto"):
            env = BatchedTaskEnv(B, A, T, device="cuda:0")
            env.load_instances(**inst)
            env.load_route_arrays(routes, route_len, member_cap=5)
            env.set_replay_placement(pl)
            for _ in range(2):
                out = env.execute_routes(reactive, fields=())
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                out = env.execute_routes(reactive, fields=())
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            row[pl] = (round(ms, 3), f"{float(out['steps'].sum()) / ms * 1e3:.3e} steps/s")
            env.close()
        print(B, "reactive" if reactive else "static", row, flush=True)
