"""-m gpu: DCM_PARAM_WIDE_MEMBERS handles (BatchedTaskEnv(member_cap=16)) -- sixteen member slots per task instead of five.

The reference's member lists are unbounded (env/task_env.py:321-322): a policy that ignores the mask (worker.py:140) can send
more agents to a task than it requires, and generate_env takes any max_coalition_size (:71).  With five slots the device freezes
such an env (DCM_FLAG_OVERFLOW); a wide handle simulates it -- checked here against the reference's own traces that run into
the five-slot limit, against the oracle under a mask-ignoring policy, and on instances with requirements up to 16 through both
the lockstep API and the persistent rollout kernel."""
import glob
import os

import numpy as np
import pytest

import helpers as H
from test_gpu_api import _anymask_action

pytestmark = pytest.mark.gpu


def test_reference_overflow_traces_run_to_the_end(gpu_device, golden_dir):
    """tests/golden/overflow_*.npz: reference episodes (mask-ignoring policy) in which a task lists a sixth member.  A five-slot
    handle stops there (test_gpu_api.py); a wide handle reproduces every decision and the terminal state bit for bit."""
    from dcmrta_amd import _lib
    from dcmrta_amd.batched_env import BatchedTaskEnv
    paths = sorted(glob.glob(os.path.join(golden_dir, "overflow_*.npz")))
    assert len(paths) >= 3
    for p in paths:
        tr = H.load_trace(p)
        A, T = int(tr["A"]), int(tr["T"])
        assert int(tr["n_members"].max()) > 5 or int(tr["overflow_step"]) >= 0
        env = BatchedTaskEnv(1, A, T, device=gpu_device, member_cap=16)
        env.load_instances(tr["depot"][None], tr["task_xy"][None], tr["req"][None], tr["dur"][None])
        g = H.run_lockstep(env, np.array([int(tr["seed_e"])], np.uint64), lambda b, i, m, l: int(tr["action"][i]))[0]
        assert g["n_steps"] == int(tr["n_steps"]) > int(tr["overflow_step"])
        for k in ("leader", "now", "mask", "agents_obs", "tasks_obs"):
            assert np.array_equal(g[k], tr[k]), (p, k)
        fin = H.gpu_final(env)[0]
        assert not fin["flags"] & (_lib.FLAG_OVERFLOW | _lib.FLAG_BAD_ACTION)
        H.assert_final_matches(fin, tr, os.path.basename(p))
        mem = env.task_members()[0].cpu().numpy()
        assert mem.shape == (T, 16) and int((mem >= 0).sum(1).max()) == int(tr["n_members"].max())
        env.close()


def test_mask_ignoring_policy_against_the_oracle(gpu_device, oracle_lib):
    """64 random instances under a policy that ignores the mask: every env whose longest member list stays within sixteen is exact
    to the end -- including the envs a five-slot handle freezes -- and the freeze happens exactly where the oracle's list exceeds 16."""
    from dcmrta_amd import _lib
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.choice import env_seeds
    from dcmrta_amd.instances import generate_batch
    B, A, T = 64, 12, 25
    inst = generate_batch(B, A, T, base_seed=400)
    seeds = env_seeds(13, 0, B)
    env = BatchedTaskEnv(B, A, T, device=gpu_device, member_cap=16).load_instances(**inst)
    got = H.run_lockstep(env, seeds, lambda b, i, m, l: _anymask_action(m, int(seeds[b]), i, T))
    fin = H.gpu_final(env)
    n_wide = n_exact = 0
    for b in range(B):
        o = oracle_lib.OracleEnv(A, T).load(inst["depot"][b], inst["task_xy"][b], inst["req"][b], inst["dur"][b])
        ref = o.rollout(int(seeds[b]), 0, oracle_lib.POLICY_ANY, cap_steps=20000)
        overflowed = bool(fin[b]["flags"] & _lib.FLAG_OVERFLOW)
        assert overflowed == (ref["max_members_seen"] > 16), b
        n = got[b]["n_steps"]
        for k in ("leader", "action", "now", "mask", "agents_obs", "tasks_obs"):
            assert np.array_equal(got[b][k], ref[k][:n]), (b, k)
        if overflowed:
            continue
        assert n == ref["n_steps"]
        H.assert_final_matches(fin[b], ref, f"env{b}")
        n_exact += 1
        n_wide += ref["max_members_seen"] > 5
    assert n_exact >= 50 and n_wide >= 5, (n_exact, n_wide)      # envs a five-slot handle would have frozen are among them
    env.close()


@pytest.mark.parametrize("A,T,rmax", [(10, 14, 8), (30, 70, 16), (24, 9, 16)])
def test_requirements_beyond_five(gpu_device, oracle_lib, A, T, rmax):
    """max_coalition_size = 8 / 16 (env/task_env.py:71 draws requirements 1..max_coalition_size): the persistent rollout kernel and
    the lockstep API against the oracle, every terminal quantity bit-exact -- incl. numpy's pairwise np.sum over 8..16 members."""
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.choice import env_seeds
    from dcmrta_amd.instances import generate_batch
    B = 24
    inst = generate_batch(B, A, T, base_seed=91)
    rng = np.random.default_rng(8)
    inst["req"] = rng.integers(1, rmax + 1, (B, T)).astype(np.int32)
    seeds = env_seeds(5, 0, B)
    refs = []
    for b in range(B):
        o = oracle_lib.OracleEnv(A, T).load(inst["depot"][b], inst["task_xy"][b], inst["req"][b], inst["dur"][b])
        refs.append(o.rollout(int(seeds[b]), 0, oracle_lib.POLICY_RANDOM, cap_steps=20000))
    assert max(int(r["n_members"].max()) for r in refs) > 5 or max(r["max_members_seen"] for r in refs) > 5
    # persistent kernel
    env = BatchedTaskEnv(B, A, T, device=gpu_device, member_cap=16).load_instances(**inst)
    with pytest.raises(Exception):
        BatchedTaskEnv(B, A, T, device=gpu_device).load_instances(**inst)          # five slots: a requirement above 5 is refused
    env.reset(seeds, observe=False)
    steps = env.rollout_random(episodes=1).cpu().numpy()
    fin = H.gpu_final(env)
    for b in range(B):
        assert steps[b] == refs[b]["n_steps"]
        H.assert_final_matches(fin[b], refs[b], f"rollout env{b}")
    # lockstep API, protocol choices, the oracle's (= the protocol's) actions
    got = H.run_lockstep(env, seeds, lambda b, i, m, l: int(refs[b]["action"][i]))
    fin = H.gpu_final(env)
    for b in range(B):
        for k in ("leader", "now", "mask", "agents_obs", "tasks_obs"):
            assert np.array_equal(got[b][k], refs[b][k]), (b, k)
        H.assert_final_matches(fin[b], refs[b], f"lockstep env{b}")
    env.close()
