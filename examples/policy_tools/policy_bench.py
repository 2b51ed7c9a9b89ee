#!/usr/bin/env python3
"""Forward time of the stand-in attention policy at rollout batch sizes (developer tool, GPU box).
    python examples/policy_tools/policy_bench.py [B A T]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dcmrta_amd.policy import AttentionNet

B, A, T = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (4096, 20, 50)))
dev = "cuda:0"
torch.manual_seed(0)
net = AttentionNet().to(dev).eval()
tasks, agents = torch.rand(B, T + 1, 5, device=dev), torch.rand(B, A, 6, device=dev)
mask = torch.rand(B, T + 1, device=dev) < 0.3
mask[:, 0] = False


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    ref = net(tasks, agents, mask)
live = ~mask
for nopad in (False, True):
    net.assume_no_padding = nopad
    for name, make in (("fp32", lambda: net), ("bf16 shadow", lambda: net.rollout_copy(torch.bfloat16)),
                       ("fp16 shadow", lambda: net.rollout_copy(torch.float16))):
        for chunk in (0,):
            m = make()

            def fwd():
                with torch.no_grad():
                    return m(tasks, agents, mask)
            try:
                ms = timeit(fwd)
                out = fwd()
                err = (out[live] - ref[live]).abs().max().item()
                agree = (out.argmax(1) == ref.argmax(1)).float().mean().item()
                print(f"no_padding={nopad} {name}: {ms:.3f} ms, max |dlogp| {err:.3e}, argmax agreement {agree:.4f}", flush=True)
            except Exception as ex:
                print(f"no_padding={nopad} {name}: FAILED {type(ex).__name__}: {str(ex)[:300]}", flush=True)
