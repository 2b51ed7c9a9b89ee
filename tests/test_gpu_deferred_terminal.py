"""dcm_step with DCM_PARAM_AUTO_RESET defers the terminal metrics of its eager steps (step_fast.hpp: the ending wave parks the final
record, k_terminal_flush computes reward + metrics later).  What callers see must not change: summary rows through dcm_summary, the
return log, episode counts -- against the oracle, episode by episode (worker.py:87,103-108; env/task_env.py:344-364,420-425)."""
import numpy as np
import pytest
import torch

import helpers as H

pytestmark = pytest.mark.gpu


def _oracle_episodes(oracle_lib, inst, seeds, A, T, b, n):
    """The first n consecutive episodes of env b under the random policy (the decision counter keeps running)."""
    out, d0 = [], 0
    o = oracle_lib.OracleEnv(A, T).load(inst["depot"][b], inst["task_xy"][b], inst["req"][b], inst["dur"][b])
    for _ in range(n):
        r = o.rollout(int(seeds[b]), d0, oracle_lib.POLICY_RANDOM, cap_steps=20000, record=False)
        out.append(r)
        d0 += r["n_steps"]
        o.clear_decisions()
    return out


@pytest.mark.parametrize("A,T,B,steps", [(20, 50, 96, 700), (12, 23, 64, 500), (64, 63, 16, 900)])
def test_summaries_read_rarely_match_the_oracle(gpu_device, oracle_lib, A, T, B, steps):
    """A collection loop that reads the summary only at the end: several periodic flushes happen on the way (every 32 steps), every
    env finishes several episodes; the return log holds EVERY episode's return, the final summary rows hold each env's last finished
    episode -- bit for bit the oracle's."""
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.choice import env_seeds
    from dcmrta_amd.instances import generate_batch
    inst = generate_batch(B, A, T, base_seed=7100 + A)
    seeds = env_seeds(23, 0, B)
    env = BatchedTaskEnv(B, A, T, device=gpu_device, auto_reset=True).load_instances(**inst)
    cap = 32
    ring = env.enable_return_log(cap)
    obs = env.reset(seeds)
    dcount = np.zeros(B, np.int64)
    for _ in range(steps):
        mk = obs.mask.cpu().numpy().astype(np.uint8)
        act = np.array([H.host_random_action(mk[b], int(seeds[b]), int(dcount[b])) for b in range(B)], np.int32)
        obs = env.step(act)
        dcount += 1
    eps = env.episodes().cpu().numpy()
    sm = env.summary().cpu().numpy()
    rl = ring.cpu().numpy()
    assert eps.min() >= 2 and eps.max() <= cap
    for b in range(B):
        ref = _oracle_episodes(oracle_lib, inst, seeds, A, T, b, int(eps[b]))
        for k, r in enumerate(ref):
            assert rl[b, k] == r["reward"], (b, k)
        last = ref[-1]
        assert sm[b, 0] == last["reward"] and int(sm[b, 1]) == int(last["finished"].sum()), b
        for i in range(6):
            assert sm[b, 2 + i] == last["metrics"][i], (b, i)
        assert sum(r["n_steps"] for r in ref) <= dcount[b]


def test_an_env_that_ends_again_before_the_flush(gpu_device, oracle_lib):
    """Tiny envs end an episode every few decisions: an env whose previous snapshot is still waiting computes its next terminal
    metrics inline (one snapshot slot per env).  Summary read every 7 steps and at the end."""
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.choice import env_seeds
    from dcmrta_amd.instances import generate_batch
    B, A, T = 48, 2, 2
    inst = generate_batch(B, A, T, base_seed=7300)
    seeds = env_seeds(29, 0, B)
    env = BatchedTaskEnv(B, A, T, device=gpu_device, auto_reset=True).load_instances(**inst)
    obs = env.reset(seeds)
    dcount = np.zeros(B, np.int64)
    refs = [_oracle_episodes(oracle_lib, inst, seeds, A, T, b, 80) for b in range(B)]
    bounds = [np.cumsum([r["n_steps"] for r in refs[b]]) for b in range(B)]
    for s in range(150):
        mk = obs.mask.cpu().numpy().astype(np.uint8)
        act = np.array([H.host_random_action(mk[b], int(seeds[b]), int(dcount[b])) for b in range(B)], np.int32)
        obs = env.step(act)
        dcount += 1
        if s % 7 == 6 or s == 149:
            eps = env.episodes().cpu().numpy()
            sm = env.summary().cpu().numpy()
            for b in range(B):
                k = int(eps[b])
                assert k == int(np.searchsorted(bounds[b], dcount[b], side="right")), (b, s)
                if k:
                    last = refs[b][k - 1]
                    assert sm[b, 0] == last["reward"], (b, s, k)
                    for i in range(6):
                        assert sm[b, 2 + i] == last["metrics"][i] or (np.isnan(sm[b, 2 + i]) and np.isnan(last["metrics"][i])), (b, s, i)
    assert int(env.episodes().min()) >= 5 and int(env.episodes().max()) >= 30


def test_capture_is_refused_while_summaries_wait_and_inline_under_capture(gpu_device):
    """A dcm_step under stream capture computes its terminal metrics inline; it refuses to be captured while rows of earlier eager
    steps may still be waiting (dcm_summary / dcm_reset clear that)."""
    from dcmrta_amd import _lib
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.choice import env_seeds
    from dcmrta_amd.instances import generate_batch
    B, A, T = 32, 6, 9
    inst = generate_batch(B, A, T, base_seed=7400)
    seeds = env_seeds(31, 0, B)
    env = BatchedTaskEnv(B, A, T, device=gpu_device, auto_reset=True).load_instances(**inst)
    ref = BatchedTaskEnv(B, A, T, device=gpu_device, auto_reset=True).load_instances(**inst)
    obs = env.reset(seeds)
    robs = ref.reset(seeds)
    act = torch.zeros(B, dtype=torch.int32, device=gpu_device)

    def policy(o):
        return torch.argmax((~o.mask).to(torch.int32), dim=1).to(torch.int32)
    for _ in range(5):
        obs = env.step(policy(obs))
        robs = ref.step(policy(robs))
    s = torch.cuda.Stream(device=gpu_device)
    s.wait_stream(torch.cuda.current_stream(gpu_device))
    torch.cuda.synchronize(gpu_device)
    g = torch.cuda.CUDAGraph()
    with pytest.raises(_lib.DcmError, match="capture"):
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            env.step(act)
    env.summary()                                           # computes what was waiting
    torch.cuda.synchronize(gpu_device)
    g = torch.cuda.CUDAGraph()
    act.copy_(policy(obs))
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        env.step(act)
    for _ in range(120):                                    # episodes end inside the replays: inline metrics
        g.replay()
        robs = ref.step(policy(robs))
        act.copy_(policy(obs))
    for _ in range(60):                                     # eager steps after the capture: inline as well (a replay could follow)
        obs = env.step(policy(obs))
        robs = ref.step(policy(robs))
        act.copy_(policy(obs))
        g.replay()
        robs = ref.step(policy(robs))
    assert int(env.episodes().min()) >= 1
    assert torch.equal(env.episodes(), ref.episodes())
    sm, rm = env.summary(), ref.summary()
    assert torch.equal(sm.view(torch.int64), rm.view(torch.int64))


def test_restart_image_follows_reloads_and_restores(gpu_device, oracle_lib):
    """The register-resident step restarts an env from a copy of the records dcm_reset produced.  The copy must follow the
    handle: new instances + reset refresh it, dcm_restore_state (records of possibly other instances) drops it -- the kernel then
    recomputes the restart -- and either way every episode is the oracle's."""
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.choice import env_seeds
    from dcmrta_amd.instances import generate_batch
    B, A, T = 40, 7, 9
    env = BatchedTaskEnv(B, A, T, device=gpu_device, auto_reset=True)

    def play(inst, seeds, steps, restore_at=None, other=None):
        ring = env.enable_return_log(64)
        env.load_instances(**inst)
        obs = env.reset(seeds)
        dcount = np.zeros(B, np.int64)
        for s in range(steps):
            if s == restore_at:                       # records of ANOTHER instance set, mid-episode, then back: the image is dropped
                keep = env.clone_state()
                env.restore_state(other)
                env.restore_state(keep)
            mk = obs.mask.cpu().numpy().astype(np.uint8)
            act = np.array([H.host_random_action(mk[b], int(seeds[b]), int(dcount[b])) for b in range(B)], np.int32)
            obs = env.step(act)
            dcount += 1
        eps = env.episodes().cpu().numpy()
        sm = env.summary().cpu().numpy()
        rl = ring.cpu().numpy()
        assert eps.min() >= 2 and eps.max() <= 64
        for b in range(B):
            ref = _oracle_episodes(oracle_lib, inst, seeds, A, T, b, int(eps[b]))
            for k, r in enumerate(ref):
                assert rl[b, k] == r["reward"], (b, k)
            for i in range(6):
                assert sm[b, 2 + i] == ref[-1]["metrics"][i], (b, i)

    inst1, inst2 = generate_batch(B, A, T, base_seed=7500), generate_batch(B, A, T, base_seed=7600)
    play(inst1, env_seeds(37, 0, B), 120)
    other = env.clone_state()                         # mid-episode records of the first instance set
    play(inst2, env_seeds(41, 0, B), 120)             # same handle, new instances: the image is refreshed by the reset
    play(inst2, env_seeds(43, 0, B), 150, restore_at=40, other=other)


def test_zero_max_time_handle_keeps_the_computed_restart(gpu_device):
    """max_time <= 0: the loop test fails before the first decision (worker.py:45), so a restart ends its episode at once -- which only
    the computed restart can account for (summary row, episode count); such a handle gets no restart image and stays consistent."""
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.choice import env_seeds
    from dcmrta_amd.instances import generate_batch
    B, A, T = 8, 5, 7
    env = BatchedTaskEnv(B, A, T, device=gpu_device, auto_reset=True, max_time=0.0).load_instances(**generate_batch(B, A, T, base_seed=7700))
    obs = env.reset(env_seeds(47, 0, B))
    for _ in range(3):
        obs = env.step(torch.zeros(B, dtype=torch.int32, device=gpu_device))
    assert not bool(obs.active.any())
    assert torch.equal(env.episodes().cpu(), torch.ones(B, dtype=torch.int32))
    assert torch.equal(env.summary()[:, 0].cpu(), torch.zeros(B, dtype=torch.float64))      # reward = -now = -0.0 (== 0.0)
