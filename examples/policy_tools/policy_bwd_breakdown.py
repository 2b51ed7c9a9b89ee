#!/usr/bin/env python3
"""Kernel-time breakdown of one REINFORCE forward + backward of the stand-in policy on a minibatch of decisions (developer tool)."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dcmrta_amd.policy import AttentionNet  # noqa: E402

N, A, T = int(sys.argv[1]) if len(sys.argv) > 1 else 8192, 20, 50
dev = "cuda:0"
torch.manual_seed(0)
net = AttentionNet().to(dev)
net.assume_no_padding = len(sys.argv) > 2 and sys.argv[2] == "nopad"
tasks, agents = torch.rand(N, T + 1, 5, device=dev), torch.rand(N, A, 6, device=dev)
mask = torch.rand(N, T + 1, device=dev) < 0.3; mask[:, 0] = False
action = torch.randint(0, T + 1, (N, 1), device=dev)
adv = torch.randn(N, 1, device=dev)


def step():
    logp = net(tasks, agents, mask)
    (-(torch.gather(logp, 1, action) * adv).sum()).backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    step()
e1.record(); torch.cuda.synchronize()
print(f"N={N} assume_no_padding={net.assume_no_padding}: {e0.elapsed_time(e1) / 5:.2f} ms per forward + backward = {N / (e0.elapsed_time(e1) / 5) * 1e3:.0f} decisions/s")
with profile(activities=[ProfilerActivity.CUDA]) as p:
    for _ in range(3):
        step()
    torch.cuda.synchronize()
rows = sorted(p.key_averages(), key=lambda r: -r.device_time_total)
tot = sum(r.device_time_total for r in rows)
for r in rows[:14]:
    print(f"{r.device_time_total / 3 / 1e3:8.3f} ms  {100 * r.device_time_total / tot:5.1f} %  x{r.count // 3:3d}  {r.key[:100]}")
