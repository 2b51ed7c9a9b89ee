"""-m gpu: the opt-in register-resident rollout kernel (dcmrta_amd/csrc/fast_rollout.hpp, DCM_FAST_ROLLOUT=1) against the oracle."""
import os

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture()
def fast_kernel():
    os.environ["DCM_FAST_ROLLOUT"] = "1"     # read by dcm_rollout_random at every call
    yield
    os.environ.pop("DCM_FAST_ROLLOUT", None)


@pytest.mark.parametrize("A,T,B", [(20, 50, 64), (5, 8, 8), (1, 1, 2), (13, 37, 8), (64, 64, 4), (33, 64, 4)])
def test_fast_rollout_matches_oracle(gpu_device, oracle_lib, fast_kernel, A, T, B):
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.choice import env_seeds
    from dcmrta_amd.instances import generate_batch
    inst = generate_batch(B, A, T, base_seed=300 + A)
    seeds = env_seeds(21, 0, B)
    env = BatchedTaskEnv(B, A, T, device=gpu_device).load_instances(**inst)
    env.reset(seeds, observe=False)
    total = np.zeros(B, np.int64)
    for ep in range(2):                                   # second call exercises the in-kernel episode restart
        total += env.rollout_random(episodes=1).cpu().numpy()
    fin = H.gpu_final(env)
    for b in range(B):
        o = oracle_lib.OracleEnv(A, T).load(inst["depot"][b], inst["task_xy"][b], inst["req"][b], inst["dur"][b])
        n1 = o.rollout(int(seeds[b]), 0, oracle_lib.POLICY_RANDOM, cap_steps=20000, record=False)["n_steps"]
        o.clear_decisions()
        ref = o.rollout(int(seeds[b]), n1, oracle_lib.POLICY_RANDOM, cap_steps=20000, record=False)
        assert total[b] == n1 + ref["n_steps"], b
        H.assert_final_matches(fin[b], ref, f"fast {A}A{T}T env{b}")


def test_fast_and_default_kernels_agree_on_observations(gpu_device, fast_kernel):
    """The observation tensors left by the last decision of an episode are identical for both kernels."""
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.choice import env_seeds
    from dcmrta_amd.instances import generate_batch
    B, A, T = 32, 20, 50
    inst = generate_batch(B, A, T, base_seed=8)
    outs = []
    for fast in (True, False):
        if fast:
            os.environ["DCM_FAST_ROLLOUT"] = "1"
        else:
            os.environ.pop("DCM_FAST_ROLLOUT", None)
        env = BatchedTaskEnv(B, A, T, device=gpu_device).load_instances(**inst)
        obs = env.reset(env_seeds(2, 0, B))
        steps = env.rollout_random(1)
        outs.append([x.clone() for x in (steps, obs.agents, obs.tasks, obs.mask, env.summary())])
    for a, b in zip(*outs):
        assert (a == b).all() or (a.is_floating_point() and ((a == b) | (a.isnan() & b.isnan())).all())


@pytest.mark.parametrize("fast", [False, True])
def test_rollout_kernels_on_tie_instances(gpu_device, oracle_lib, golden_dir, fast):
    """Symmetric instances (tests/golden/micro_*.npz) produce events with several groups at different locations; both
    persistent kernels must order them like np.unique(axis=0) under the random policy too (checked against the oracle)."""
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.choice import env_seeds
    if fast:
        os.environ["DCM_FAST_ROLLOUT"] = "1"
    try:
        for name in ("micro_ties4", "micro_ties2y", "micro_ties_mixed"):
            tr = H.load_trace(os.path.join(golden_dir, name + ".npz"))
            A, T, B = int(tr["A"]), int(tr["T"]), 16
            env = BatchedTaskEnv(B, A, T, device=gpu_device)
            env.load_instances(np.repeat(tr["depot"][None], B, 0), np.repeat(tr["task_xy"][None], B, 0),
                               np.repeat(tr["req"][None], B, 0), np.repeat(tr["dur"][None], B, 0))
            seeds = env_seeds(77, 0, B)
            env.reset(seeds, observe=False)
            steps = env.rollout_random(1).cpu().numpy()
            fin = H.gpu_final(env)
            for b in range(B):
                o = oracle_lib.OracleEnv(A, T).load(tr["depot"], tr["task_xy"], tr["req"], tr["dur"])
                ref = o.rollout(int(seeds[b]), 0, oracle_lib.POLICY_RANDOM, record=False)
                assert steps[b] == ref["n_steps"], (name, b)
                H.assert_final_matches(fin[b], ref, f"{name} env{b} fast={fast}")
    finally:
        os.environ.pop("DCM_FAST_ROLLOUT", None)
