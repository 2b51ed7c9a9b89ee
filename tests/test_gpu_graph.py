"""-m gpu: the HIP-graph captured decision loop (dcmrta_amd/graph_rollout.py) plays episodes of the reference env: the
decisions it records replay bit-exactly through the oracle, also after the batch changes between uniform and ragged."""
import numpy as np
import pytest
import torch

from test_gpu_runner import _replay_recorded

pytestmark = pytest.mark.gpu


def first_valid(obs):
    return torch.argmax((~obs.mask).to(torch.int32), dim=1)   # lowest unmasked action id


def test_graph_replay_matches_oracle(gpu_device, oracle_lib):
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.choice import env_seeds
    from dcmrta_amd.graph_rollout import GraphedRollout
    from dcmrta_amd.instances import generate_batch, generate_batch_ranges
    B, A, T = 48, 20, 50
    inst = generate_batch(B, A, T, base_seed=21)
    seeds = env_seeds(4, 0, B)
    env = BatchedTaskEnv(B, A, T, device=gpu_device).load_instances(**inst)
    g = GraphedRollout(env, first_valid, check_every=4, record=True)
    summary, n = g.run(seeds)
    rec = {k: v[:n] for k, v in g.rec.items()}
    assert int(rec["active"].sum()) == int(env.status()["decisions"].sum())
    _replay_recorded(oracle_lib, rec, summary.cpu().numpy(), inst, seeds, A, T)
    # the captured graph is reusable: other instances and seeds, same graph object
    inst2, seeds2 = generate_batch(B, A, T, base_seed=99), env_seeds(5, 0, B)
    env.load_instances(**inst2)
    graph_before = g.graph
    summary2, n2 = g.run(seeds2)
    assert g.graph is graph_before
    _replay_recorded(oracle_lib, {k: v[:n2] for k, v in g.rec.items()}, summary2.cpu().numpy(), inst2, seeds2, A, T)
    # switching the handle to a ragged batch changes what the captured dcm_step has baked in (per-env sizes pointer, kernel
    # instantiation): the rollout must re-capture instead of replaying a stale graph
    rag = generate_batch_ranges(range(700, 700 + B), (10, 20), (20, 50))
    env.load_instances(**rag)
    summary3, n3 = g.run(seeds)
    assert g.graph is not graph_before
    _replay_recorded(oracle_lib, {k: v[:n3] for k, v in g.rec.items()}, summary3.cpu().numpy(), rag, seeds, A, T,
                     n_agents=rag["n_agents"], n_tasks=rag["n_tasks"])
    # ... and back to uniform
    env.load_instances(**inst)
    summary4, n4 = g.run(seeds)
    assert torch.equal(summary4, summary)


def test_record_capacity_is_enforced(gpu_device):
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.choice import env_seeds
    from dcmrta_amd.graph_rollout import GraphedRollout
    from dcmrta_amd.instances import generate_batch
    B, A, T = 8, 10, 20
    env = BatchedTaskEnv(B, A, T, device=gpu_device).load_instances(**generate_batch(B, A, T, base_seed=3))
    g = GraphedRollout(env, first_valid, check_every=4, record=True, capacity=8)
    with pytest.raises(RuntimeError, match="capacity"):
        g.run(env_seeds(1, 0, B))
