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


def test_compacted_policy_buckets_match_oracle(gpu_device, oracle_lib):
    """buckets: the policy runs only on the envs still active (gathered rows, one graph per bucket size).  Same episodes as
    the full-batch loop, and they replay through the oracle; the small buckets really are used once envs finish."""
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.choice import env_seeds
    from dcmrta_amd.graph_rollout import GraphedRollout
    from dcmrta_amd.instances import generate_batch
    from dcmrta_amd.instances import generate_batch_ranges
    B, A, T = 64, 20, 50
    inst = generate_batch_ranges(range(900, 900 + B), (4, 20), (5, 50))       # very different sizes -> very different lengths
    seeds = env_seeds(9, 0, B)
    env = BatchedTaskEnv(B, A, T, device=gpu_device).load_instances(**inst)
    full = GraphedRollout(env, first_valid, check_every=4, record=True)
    s_full, n_full = full.run(seeds)
    rec_full = {k: v[:n_full].clone() for k, v in full.rec.items()}
    g = GraphedRollout(env, first_valid, check_every=4, record=True, buckets=(1.0, 0.5, 0.25, 0.125))
    s_b, n_b = g.run(seeds)
    assert torch.equal(s_b, s_full) and n_b == n_full
    rec = {k: v[:n_b] for k, v in g.rec.items()}
    act = rec["active"]
    assert torch.equal(act, rec_full["active"])
    for k in ("agents", "tasks", "mask", "leader"):
        assert torch.equal(rec[k][act], rec_full[k][act]), k
    assert torch.equal(rec["action"][act], rec_full["action"][act])
    assert g.bucket_steps[B] > 0 and sum(v for n, v in g.bucket_steps.items() if n < B) > 0, g.bucket_steps
    _replay_recorded(oracle_lib, rec, s_b.cpu().numpy(), inst, seeds, A, T, n_agents=inst["n_agents"], n_tasks=inst["n_tasks"])
    # with auto-reset and an episode limit the active set also only shrinks: 2 episodes per env, compacted
    env2 = BatchedTaskEnv(B, A, T, device=gpu_device, auto_reset=True, auto_reset_episodes=2).load_instances(**inst)
    g2 = GraphedRollout(env2, first_valid, check_every=4, buckets=(1.0, 0.5, 0.25))
    g2.run(seeds)
    assert (env2.episodes() == 2).all()
    ref_env = BatchedTaskEnv(B, A, T, device=gpu_device, auto_reset=True, auto_reset_episodes=2).load_instances(**inst)
    GraphedRollout(ref_env, first_valid, check_every=4).run(seeds)
    assert torch.equal(env2.summary(), ref_env.summary()) and torch.equal(env2.status()["decisions"], ref_env.status()["decisions"])


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


def test_attention_policy_graph_loop_at_full_batch(gpu_device, oracle_lib):
    """BASELINE configs[2] at its real size: GraphedRollout + the attention net (fp32, the reference's arithmetic) + compaction
    buckets at B = 4096 for one sampled episode per env -- the size at which two ROCm graph hazards were hit (torch.multinomial
    faulting on replay, concurrent replays hanging).  A 32-env sample of the recorded episodes replays through the
    oracle; the record's window rounding lets the last episode end in the final partial check window without a false overflow."""
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.choice import env_seeds
    from dcmrta_amd.graph_rollout import GraphedRollout
    from dcmrta_amd.instances import generate_batch
    from dcmrta_amd.policy import AttentionNet
    B, A, T = 4096, 20, 50
    torch.manual_seed(3)
    net = AttentionNet().to(gpu_device).eval()
    net.assume_no_padding = True
    inst = generate_batch(B, A, T, base_seed=0)
    seeds = env_seeds(0, 0, B)
    env = BatchedTaskEnv(B, A, T, device=gpu_device).load_instances(**inst)

    @torch.no_grad()
    def policy(ob):
        lp = net(ob.tasks, ob.agents, ob.mask)
        return torch.argmax(lp - torch.empty_like(lp).exponential_(1.0).log(), dim=1).to(torch.int32)
    g = GraphedRollout(env, policy, check_every=4, record=True, capacity=250, buckets=(1.0, 0.5, 0.25, 0.125))
    assert g.capacity == 252                                     # rounded up to whole check windows
    summary, n = g.run(seeds)
    flags = env.status()["flags"].cpu().numpy()
    assert not (flags & 0x138).any() and (flags & 1).all()       # nobody frozen, everybody done
    assert sum(v for k, v in g.bucket_steps.items() if k < B) > 0
    dec = env.status()["decisions"]
    assert int(g.rec["active"][:n].sum()) == int(dec.sum()) and 60 * B < int(dec.sum()) < 200 * B
    pick = np.linspace(0, B - 1, 32).astype(int)
    rec = {k: v[:n][:, pick].contiguous() for k, v in g.rec.items()}
    sub = {k: v[pick] for k, v in inst.items()}
    _replay_recorded(oracle_lib, rec, summary[pick].cpu().numpy(), sub, seeds[pick], A, T)
    assert not rec["mask"].gather(2, rec["action"].unsqueeze(2))[rec["active"]].any()     # no sampled action was masked
