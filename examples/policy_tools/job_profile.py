#!/usr/bin/env python3
"""Where a BatchedRunner.job spends its wall time (developer tool, GPU box): cProfile by cumulative time + GPU busy share."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dcmrta_amd.runner import BatchedRunner  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
torch.manual_seed(0)
r = BatchedRunner(n_envs=B, device="cuda:0")
w = {k: v.clone() for k, v in r.get_weights().items()}
r.job(w, w, 0, 20, 50)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
res, metrics, info = r.job(w, w, 1, 20, 50)
torch.cuda.synchronize()
pr.disable()
print(f"job wall {time.perf_counter() - t0:.3f} s, {res[0].shape[0]} recorded decisions, sampled steps {r.last['n_steps']}")
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
