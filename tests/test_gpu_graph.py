"""-m gpu: the HIP-graph captured decision loop plays the same episodes as the eager loop."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def first_valid(obs):
    return torch.argmax((~obs.mask).to(torch.int32), dim=1)   # lowest unmasked action id


def test_graph_replay_equals_eager(gpu_device):
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.choice import env_seeds
    from dcmrta_amd.graph_rollout import GraphedRollout
    from dcmrta_amd.instances import generate_batch
    B, A, T = 128, 20, 50
    inst = generate_batch(B, A, T, base_seed=21)
    seeds = env_seeds(4, 0, B)
    env = BatchedTaskEnv(B, A, T, device=gpu_device).load_instances(**inst)
    obs = env.reset(seeds)
    steps = 0
    while bool(obs.active.any()):
        obs = env.step(first_valid(obs).int())
        steps += 1
    eager = env.summary().clone()
    dec_eager = env.status()["decisions"].clone()
    g = GraphedRollout(env, first_valid, check_every=4)
    graphed, n = g.run(seeds)
    assert n >= steps and torch.equal(graphed, eager)
    assert torch.equal(env.status()["decisions"], dec_eager)
    again, _ = g.run(seeds)                                   # the captured graph is reusable
    assert torch.equal(again, eager)
