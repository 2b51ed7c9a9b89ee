#!/usr/bin/env python3
"""Kernel-time breakdown of one forward of the stand-in attention policy at rollout batch size (developer tool, GPU box).
    python examples/policy_tools/policy_breakdown.py [fp32|fp16|bf16] [B]"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dcmrta_amd.policy import AttentionNet  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
B, A, T = int(sys.argv[2]) if len(sys.argv) > 2 else 4096, 20, 50
dev = "cuda:0"
torch.manual_seed(0)
net = AttentionNet().to(dev).eval(); net.assume_no_padding = True
tasks, agents = torch.rand(B, T + 1, 5, device=dev), torch.rand(B, A, 6, device=dev)
mask = torch.rand(B, T + 1, device=dev) < 0.3; mask[:, 0] = False
m = net if prec == "fp32" else net.rollout_copy({"bf16": torch.bfloat16, "fp16": torch.float16}[prec])
with torch.no_grad():
    for _ in range(5):
        m(tasks, agents, mask)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        m(tasks, agents, mask)
    e1.record(); torch.cuda.synchronize()
    print(f"{prec} B={B}: {e0.elapsed_time(e1) / 10:.3f} ms per forward (eager)")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = m(tasks, agents, mask)
    g.replay(); torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"{prec} B={B}: {e0.elapsed_time(e1) / 10:.3f} ms per forward (HIP graph)")
    with profile(activities=[ProfilerActivity.CUDA]) as p:
        for _ in range(5):
            m(tasks, agents, mask)
        torch.cuda.synchronize()
rows = sorted(p.key_averages(), key=lambda r: -r.device_time_total)
tot = sum(r.device_time_total for r in rows)
for r in rows[:18]:
    print(f"{r.device_time_total / 5 / 1e3:8.3f} ms  {100 * r.device_time_total / tot:5.1f} %  x{r.count // 5:3d}  {r.key[:110]}")
print(f"{tot / 5 / 1e3:8.3f} ms total device time per forward")
