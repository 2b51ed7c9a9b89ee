#!/usr/bin/env python3
"""Fixture for the policy tensor contract: a seeded small-width reference AttentionNet
(embedding_dim=16), its parameters, a batch of inputs (one row-padded) and the reference outputs.
Run in the build container only:  PYTHONPATH=/root/reference python3 tests/golden/make_policy_golden.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.environ.get("DCMRTA_REFERENCE", "/root/reference"))
sys.dont_write_bytecode = True
from attention import AttentionNet  # noqa: E402 (reference, read-only)

torch.manual_seed(0)
net = AttentionNet(6, 5, 16).eval()
B, A, T = 3, 7, 11
tasks = torch.randn(B, T + 1, 5)
agents = torch.randn(B, A, 6)
tasks[1, -2:, :] = -1   # padded task rows (attention.py:10-18)
agents[2, -1:, :] = -1  # padded agent row
mask = torch.rand(B, T + 1) < 0.4
mask[:, 0] = False
mask[1, -2:] = True
with torch.no_grad():
    logp = net(tasks, agents, mask)
out = {"in_tasks": tasks.numpy(), "in_agents": agents.numpy(), "in_mask": mask.numpy(), "out_logp": logp.numpy()}
for k, v in net.state_dict().items():
    out["sd." + k] = v.numpy()
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "policy_kat.npz"), **out)
print("logp", logp[0, :4])
