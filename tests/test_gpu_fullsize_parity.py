"""-m gpu: oracle parity AT THE SIZES THE BENCH NUMBERS ARE QUOTED ON (worker.py:45-87,103-108 for every env of the batch).

The other suites compare at most a few dozen envs per case; batch-size-dependent code paths (scratch placement, staged observation
stores at >= 8192 envs, the replay's LDS / HBM placement, sub-batch streams) only exist at the bench sizes.  Here the exact
bench.py workloads -- same instances (generate_batch(base_seed=0, first=...)), same seeds (env_seeds(0, first, B)), same call
sequence -- are compared with the multi-threaded oracle for EVERY env: per-env decision totals, every episode's return, and the
last episode's finished-task count + six perf metrics, all bit-exact."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
THREADS = max(1, min(32, len(os.sched_getaffinity(0))))


def _assert_batch_equal(got, ref, name):
    """got / ref: dicts of numpy arrays with the same keys; floating-point arrays are compared by bit pattern."""
    for k, r in ref.items():
        g = np.asarray(got[k])
        r = np.asarray(r)
        assert g.shape == r.shape, (name, k, g.shape, r.shape)
        same = (g.view(np.uint64) == r.view(np.uint64)) if r.dtype == np.float64 else (g.astype(np.int64) == r.astype(np.int64))
        if not same.all():
            bad = np.argwhere(~same)
            raise AssertionError(f"{name}: {k} differs for {len(bad)} of {same.size} entries; first {bad[0].tolist()}: "
                                 f"got {g[tuple(bad[0])]!r}, oracle {r[tuple(bad[0])]!r}")


def _rollout_pass(dev, A, T, first, B, episodes, stream=None):
    """What bench.py's SubBatch does for one pass, on a fresh handle: returns numpy results of the block [first, first + B)."""
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.choice import env_seeds
    from dcmrta_amd.instances import generate_batch
    inst = generate_batch(B, A, T, base_seed=0, first=first)
    seeds = env_seeds(0, first, B)
    env = BatchedTaskEnv(B, A, T, device=dev).load_instances(**inst)
    ring = env.enable_return_log(episodes)
    env.reset(seeds, observe=False)
    return env, ring, inst, seeds


def _collect(env, ring, steps):
    sm = env.summary().cpu().numpy()
    flags = env.status()["flags"].cpu().numpy()
    assert (flags & 0x13C).sum() == 0, "env error flags set"
    return dict(steps=steps.cpu().numpy(), returns=ring.cpu().numpy(), reward=sm[:, 0].copy(), n_finished=sm[:, 1].astype(np.int32),
                metrics=np.ascontiguousarray(sm[:, 2:8]))


def _oracle_rollout(oracle_lib, inst, seeds, A, episodes):
    r = oracle_lib.batch_rollout_full(inst["depot"], inst["task_xy"], inst["req"], inst["dur"], seeds, A, episodes=episodes,
                                      threads=THREADS)
    return {k: r[k] for k in ("steps", "returns", "reward", "n_finished", "metrics")}


@pytest.mark.parametrize("streams", [1, 4])
def test_config2_every_env_against_the_oracle(gpu_device, oracle_lib, streams):
    """BASELINE configs[1] exactly as bench.py runs it: 4096 envs x 20A/50T, three episodes per env in one persistent launch,
    once as one 4096-env launch and once as four 1024-env sub-batches on four HIP streams."""
    A, T, B, EP = 20, 50, 4096, 3
    dev = torch.device(gpu_device)
    side = [torch.cuda.Stream(device=dev) for _ in range(streams)]
    subs = []
    for k in range(streams):
        lo, hi = k * B // streams, (k + 1) * B // streams
        subs.append((lo,) + _rollout_pass(gpu_device, A, T, lo, hi - lo, EP))
    torch.cuda.synchronize(dev)
    steps = []
    for k, (lo, env, ring, inst, seeds) in enumerate(subs):
        if streams == 1:
            steps.append(env.rollout_random(episodes=EP))
        else:
            side[k].wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side[k]):
                steps.append(env.rollout_random(episodes=EP))
    torch.cuda.synchronize(dev)
    total = 0
    for (lo, env, ring, inst, seeds), st in zip(subs, steps):
        got = _collect(env, ring, st)
        _assert_batch_equal(got, _oracle_rollout(oracle_lib, inst, seeds, A, EP), f"config 2, {streams} stream(s), block at {lo}")
        total += int(got["steps"].sum())
        env.close()
    assert total > 1_400_000          # the 1.47e6 decisions of a bench pass


def test_config4_shard_every_env_against_the_oracle(gpu_device, oracle_lib):
    """The per-GPU shard of BASELINE configs[3] on 8 GPUs: 8192 envs x 50A/200T, one episode (k_rollout_fast_mc at the grid size
    where its scratch / LDS placement choices are the bench's)."""
    A, T, B = 50, 200, 8192
    env, ring, inst, seeds = _rollout_pass(gpu_device, A, T, 0, B, 1)
    got = _collect(env, ring, env.rollout_random(episodes=1))
    _assert_batch_equal(got, _oracle_rollout(oracle_lib, inst, seeds, A, 1), "config-4 shard")
    # a second pass continues the decision counter: the oracle's second episode
    got2 = _collect(env, ring, env.rollout_random(episodes=1))
    ref2 = oracle_lib.batch_rollout_full(inst["depot"][:512], inst["task_xy"][:512], inst["req"][:512], inst["dur"][:512], seeds[:512], A,
                                         episodes=2, threads=THREADS)
    assert np.array_equal(got2["returns"][:512, 0], ref2["returns"][:, 1])
    assert np.array_equal(got["steps"][:512] + got2["steps"][:512], ref2["steps"])
    env.close()


def test_midsize_batch_every_env_against_the_oracle(gpu_device, oracle_lib):
    """bench.py's mid-size row: 4096 envs x 70A/130T, three episodes per env (k_rollout_fast_g)."""
    A, T, B, EP = 70, 130, 4096, 3
    env, ring, inst, seeds = _rollout_pass(gpu_device, A, T, 0, B, EP)
    got = _collect(env, ring, env.rollout_random(episodes=EP))
    _assert_batch_equal(got, _oracle_rollout(oracle_lib, inst, seeds, A, EP), "70A/130T batch")
    env.close()


@pytest.mark.parametrize("placement", ["hbm", "lds"])
def test_config5_shard_against_the_oracle(gpu_device, oracle_lib, placement):
    """The per-GPU shard of BASELINE configs[4]: 100A/500T route replay with dynamic arrivals at the reference's constants
    (env/task_env.py:562-599), synthetic routes as in bench.py, replay scratch in HBM (the automatic choice at this batch: all
    8192 envs) and in LDS (2048 envs)."""
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.instances import generate_batch, synthetic_route_arrays
    A, T = 100, 500
    B = 8192 if placement == "hbm" else 2048
    inst = generate_batch(B, A, T, base_seed=0, first=0)
    routes, route_len = synthetic_route_arrays(inst["req"], A, max_task=100)
    env = BatchedTaskEnv(B, A, T, device=gpu_device).load_instances(**inst)
    env.set_visibility(20, 20, 10, 100)
    env.load_route_arrays(routes, route_len, member_cap=5)
    env.set_replay_placement(placement)
    out = env.execute_routes(True, fields=())
    flags = out["flags"].cpu().numpy()
    sm = out["summary"].cpu().numpy()
    ref = oracle_lib.batch_replay(inst["depot"], inst["task_xy"], inst["req"], inst["dur"], routes, route_len, reactive=True,
                                  visibility=(20, 20, 10, 100), threads=THREADS)
    assert not (flags & 0x58).any()
    assert np.array_equal((flags & 4) != 0, ref["status"] == 1)        # both end by the zero-decider guard (tasks 101.. never appear)
    got = dict(steps=out["steps"].cpu().numpy(), reward=sm[:, 0].copy(), n_finished=sm[:, 1].astype(np.int32),
               metrics=np.ascontiguousarray(sm[:, 2:8]))
    _assert_batch_equal(got, {k: ref[k] for k in got}, f"config-5 shard, scratch in {placement}")
    assert int(got["steps"].sum()) > 500 * B
    env.close()


def test_lockstep_65536_envs_sampled_against_the_oracle(gpu_device, oracle_lib):
    """The lockstep kernel at the batch its HBM roofline is quoted on: 65 536 envs x 20A/50T through 30 dcm_step calls under a
    random valid policy; every 512th env's whole trace -- leader, mask and both observation tensors at each of the 30 decisions,
    event times -- is replayed through the oracle with the same actions (worker.py:54-84)."""
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.choice import env_seeds
    from dcmrta_amd.instances import generate_batch
    A, T, B, N = 20, 50, 65536, 30
    inst = generate_batch(B, A, T, base_seed=0)
    seeds = env_seeds(0, 0, B)
    env = BatchedTaskEnv(B, A, T, device=gpu_device).load_instances(**inst)
    sample = torch.arange(0, B, 512, device=gpu_device)
    gen = torch.Generator(device=gpu_device).manual_seed(5)
    obs = env.reset(seeds)
    rec = dict(leader=[], mask=[], agents=[], tasks=[], action=[], now=[])
    for _ in range(N):
        assert bool(obs.active.all())           # 30 decisions into a 20A/50T episode no env is over yet
        act = torch.multinomial((~obs.mask).float(), 1, generator=gen).squeeze(1).int()
        rec["leader"].append(obs.leader[sample].cpu().numpy())
        rec["mask"].append(obs.mask[sample].cpu().numpy().astype(np.uint8))
        rec["agents"].append(obs.agents[sample].cpu().numpy())
        rec["tasks"].append(obs.tasks[sample].cpu().numpy())
        rec["action"].append(act[sample].cpu().numpy())
        rec["now"].append(env.status()["now"][sample].cpu().numpy())
        obs = env.step(act)
    rec = {k: np.stack(v, axis=1) for k, v in rec.items()}       # [sample, N, ...]
    for i, b in enumerate(sample.cpu().numpy()):
        o = oracle_lib.OracleEnv(A, T).load(inst["depot"][b], inst["task_xy"][b], inst["req"][b], inst["dur"][b])
        ref = o.rollout(int(seeds[b]), 0, oracle_lib.POLICY_INJECTED, cap_steps=N, inj_action=rec["action"][i], allow_cap=True)
        assert ref["n_steps"] == N
        assert np.array_equal(ref["leader"], rec["leader"][i]), b
        assert np.array_equal(ref["mask"], rec["mask"][i]), b
        assert np.array_equal(ref["agents_obs"].view(np.uint32), rec["agents"][i].view(np.uint32)), b
        assert np.array_equal(ref["tasks_obs"].view(np.uint32), rec["tasks"][i].view(np.uint32)), b
        assert np.array_equal(ref["now"], rec["now"][i]), b
    env.close()
