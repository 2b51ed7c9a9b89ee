#!/usr/bin/env python3
"""Does evaluating the gated feed-forward in row chunks (intermediates stay in the 256 MB MALL) beat one pass?  (developer probe)"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dev = "cuda:0"
rows, D, H = 4096 * 51, 128, 512
dt = {"fp32": torch.float32, "fp16": torch.float16}[sys.argv[1] if len(sys.argv) > 1 else "fp32"]
x = torch.randn(rows, D, device=dev, dtype=dt)
VW = torch.randn(2 * H, D, device=dev, dtype=dt) * 0.05
W2 = torch.randn(D, H, device=dev, dtype=dt) * 0.05


def full(x):
    return torch.addmm(x, F.glu(x @ VW.t(), dim=-1), W2.t())


def chunked(x, n):
    out = torch.empty_like(x)
    for lo in range(0, x.shape[0], n):
        xs = x[lo:lo + n]
        torch.addmm(xs, F.glu(xs @ VW.t(), dim=-1), W2.t(), out=out[lo:lo + n])
    return out


def bench(f, *a):
    for _ in range(3):
        f(*a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f(*a)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20


with torch.no_grad():
    ref = full(x)
    print(f"{dt} full: {bench(full, x):.3f} ms")
    for n in (8192, 16384, 32768, 65536):
        y = chunked(x, n)
        print(f"chunk {n}: {bench(chunked, x, n):.3f} ms  max|diff| {float((y - ref).abs().max()):.2e}")
