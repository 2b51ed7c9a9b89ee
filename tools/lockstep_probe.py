#!/usr/bin/env python3
"""k_step (lockstep API) throughput probe: B envs, N batched steps, device-side random policy, no host sync in the loop."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dcmrta_amd.batched_env import BatchedTaskEnv
from dcmrta_amd.choice import env_seeds
from dcmrta_amd.instances import generate_batch
from dcmrta_amd.roofline import algorithmic_bytes_per_step

from dcmrta_amd import _lib
print("build_id", _lib.build_id())
B, A, T, N = (int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (65536, 20, 50, 60)))
# `steady [skip]`: DCM_PARAM_AUTO_RESET on and the first `skip` (default 150) steps untimed -- the steady state of a collection loop,
# where every launch of a machine-sized batch holds a few envs whose episode ends in it (terminal metrics + restart: the slowest
# waves of the launch).  Without it the window is the first N decisions of an episode, where no env ends.
STEADY = len(sys.argv) > 5 and sys.argv[5] == "steady"
SKIP = int(sys.argv[6]) if STEADY and len(sys.argv) > 6 else (150 if STEADY else 0)
env = BatchedTaskEnv(B, A, T, auto_reset=STEADY).load_instances(**generate_batch(B, A, T, 0))
obs = env.reset(env_seeds(0, 0, B))
for _ in range(SKIP):
    obs = env.step(torch.multinomial((~obs.mask).float(), 1).squeeze(1).int())
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(N)]
for i in range(N):
    act = torch.multinomial((~obs.mask).float(), 1).squeeze(1).int()
    ev[i][0].record()
    obs = env.step(act)
    ev[i][1].record()
torch.cuda.synchronize()
ms = sorted(a.elapsed_time(b) for a, b in ev)
med = ms[len(ms) // 2]
W = algorithmic_bytes_per_step(A, T)
print(f"B={B} {A}A/{T}T{' steady state' if STEADY else ''} k_step min {ms[0]*1e3:.1f} p25 {ms[len(ms)//4]*1e3:.1f} max {ms[-1]*1e3:.1f} mean {sum(ms)/len(ms)*1e3:.1f} us")
print(f"B={B} {A}A/{T}T k_step median {med*1e3:.1f} us  -> {B/med*1e3:.3e} steps/s, {B*W/med/1e6:.0f} GB/s algorithmic "
      f"({B*W/med/1e6/8000*100:.1f} % of 8 TB/s)")
