"""-m gpu: the HIP path (through the C ABI) against the committed reference goldens and the oracle."""
import json
import os

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def _env(B, A, T, dev, **kw):
    from dcmrta_amd.batched_env import BatchedTaskEnv
    return BatchedTaskEnv(B, A, T, device=dev, **kw)


def test_native_library_loaded(gpu_device):
    from dcmrta_amd import _lib
    lib = _lib.load()
    assert lib.dcm_abi_version() == _lib.ABI_VERSION == 5
    with open("/proc/self/maps") as f:
        assert "libdcmrta_hip.so" in f.read()


def test_distance_known_answers(gpu_device, golden_dir):
    """G5: np.linalg.norm of the reference (env/task_env.py:161-163) == device sqrt(fma(dy,dy,dx*dx)); d/0.2 exact."""
    from dcmrta_amd.batched_env import device_distance
    z = np.load(os.path.join(golden_dir, "distance_kat.npz"))
    d, t = device_distance(z["a"], z["b"], gpu_device)
    assert np.array_equal(d, z["dist"])
    assert np.array_equal(t, z["dist"] / 0.2)


def test_travel_time_division_is_ieee(gpu_device):
    """travel_time = d / 0.2 (env/task_env.py:315) is computed as Markstein's 3-instruction division by a constant
    (csrc/common.hpp over_velocity); it must be the correctly rounded IEEE quotient for every distance: 2^22 random
    mantissas over 600 binades + the unit-square range the instances live in."""
    from dcmrta_amd.batched_env import device_distance
    rng = np.random.default_rng(3)
    n = 1 << 21
    wide = np.ldexp(1.0 + rng.random(n), rng.integers(-300, 300, n))
    unit = rng.random(n) * 1.5
    x = np.concatenate([wide, unit, [0.0, 0.2, 1.0, 5.0, 2.0 ** -400, 2.0 ** 400]])   # x*x must stay finite and normal
    a = np.stack([x, np.zeros_like(x)], 1)
    d, t = device_distance(a, np.zeros_like(a), gpu_device)
    assert np.array_equal(d, x)
    assert np.array_equal(t, x / 0.2)


def _groups():
    g = {}
    for p in H.full_traces():
        tr = H.load_trace(p)
        g.setdefault((int(tr["A"]), int(tr["T"])), []).append((os.path.basename(p), tr))
    return g


@pytest.mark.parametrize("shape", sorted(_groups().keys()))
def test_golden_traces_injected(gpu_device, shape):
    """G4 full traces replayed with injected (leader, followers, action): every observation tensor, mask, event
    time and terminal quantity must equal what the reference produced."""
    A, T = shape
    traces = _groups()[shape]
    B = len(traces)
    env = _env(B, A, T, gpu_device)
    env.load_instances(np.stack([t["depot"] for _, t in traces]), np.stack([t["task_xy"] for _, t in traces]),
                       np.stack([t["req"] for _, t in traces]), np.stack([t["dur"] for _, t in traces]))
    seeds = np.array([int(t["seed_e"]) for _, t in traces], np.uint64)
    inject = dict(leader=[t["leader"] for _, t in traces], nfol=[t["nfol"] for _, t in traces],
                  followers=[t["followers"] for _, t in traces])
    got = H.run_lockstep(env, seeds, lambda b, i, m, l: int(traces[b][1]["action"][i]), inject=inject)
    fin = H.gpu_final(env)
    for b, (name, tr) in enumerate(traces):
        g = got[b]
        assert g["n_steps"] == int(tr["n_steps"]), name
        for k in ("leader", "now", "mask", "agents_obs", "tasks_obs"):
            assert np.array_equal(g[k], tr[k]), f"{name}: {k} differs at step {_first_diff(g[k], tr[k])}"
        assert fin[b]["flags"] & 1
        H.assert_final_matches(fin[b], tr, name)


def _first_diff(a, b):
    for i in range(min(len(a), len(b))):
        if not np.array_equal(a[i], b[i]):
            return i
    return min(len(a), len(b))


@pytest.mark.parametrize("shape", sorted(_groups().keys()))
def test_golden_traces_protocol(gpu_device, shape):
    """Same traces, but leader and followers are DRAWN on the device from the keyed choice protocol: for the
    random-policy traces the host supplies the slot-1 action, for the others the recorded action."""
    A, T = shape
    traces = _groups()[shape]
    env = _env(len(traces), A, T, gpu_device)
    env.load_instances(np.stack([t["depot"] for _, t in traces]), np.stack([t["task_xy"] for _, t in traces]),
                       np.stack([t["req"] for _, t in traces]), np.stack([t["dur"] for _, t in traces]))
    seeds = np.array([int(t["seed_e"]) for _, t in traces], np.uint64)

    def policy(b, i, mask, leader):
        name, tr = traces[b]
        if "_random_" in name:
            return H.host_random_action(mask, int(tr["seed_e"]), i)
        return int(tr["action"][i])   # nearest / first-valid / micro scenarios: recorded actions

    got = H.run_lockstep(env, seeds, policy)
    fin = H.gpu_final(env)
    for b, (name, tr) in enumerate(traces):
        g = got[b]
        assert g["n_steps"] == int(tr["n_steps"]), name
        for k in ("leader", "action", "now", "mask", "agents_obs", "tasks_obs"):
            assert np.array_equal(g[k], tr[k]), f"{name}: {k} differs at step {_first_diff(g[k], tr[k])}"
        H.assert_final_matches(fin[b], tr, name)


def test_rollout_kernel_matches_oracle_20A50T(gpu_device, oracle_lib):
    """Persistent random-policy rollout kernel (config-2 path) vs the oracle on the same instances and seeds."""
    from dcmrta_amd.choice import env_seeds
    from dcmrta_amd.instances import generate_batch
    B, A, T = 96, 20, 50
    inst = generate_batch(B, A, T, base_seed=100)
    seeds = env_seeds(77, 0, B)
    env = _env(B, A, T, gpu_device)
    env.load_instances(**inst)
    env.reset(seeds, observe=False)
    steps = env.rollout_random(episodes=1).cpu().numpy()
    fin = H.gpu_final(env)
    for b in range(B):
        o = oracle_lib.OracleEnv(A, T).load(inst["depot"][b], inst["task_xy"][b], inst["req"][b], inst["dur"][b])
        ref = o.rollout(int(seeds[b]), 0, oracle_lib.POLICY_RANDOM, record=False)
        assert steps[b] == ref["n_steps"], b
        H.assert_final_matches(fin[b], ref, f"env{b}")


@pytest.mark.parametrize("A,T,B", [(5, 8, 8), (1, 1, 4), (13, 37, 8), (64, 64, 4), (21, 51, 6), (70, 130, 4), (50, 200, 6), (100, 500, 2),
                                   (128, 1023, 2)])
def test_rollout_kernel_matches_oracle_shapes(gpu_device, oracle_lib, A, T, B):
    """Edge shapes: 1 agent 1 task, exactly one wave of agents, the <64,64> layout class, A > 64 (two mask words), BASELINE
    configs 4-5 sizes, and the maximum the ABI accepts (128 agents, 1023 tasks: a 104 KB record, one workgroup per CU)."""
    from dcmrta_amd.choice import env_seeds
    from dcmrta_amd.instances import generate_batch
    inst = generate_batch(B, A, T, base_seed=500 + A)
    seeds = env_seeds(9, 0, B)
    env = _env(B, A, T, gpu_device)
    env.load_instances(**inst)
    env.reset(seeds, observe=False)
    steps = env.rollout_random(episodes=1).cpu().numpy()
    fin = H.gpu_final(env)
    for b in range(B):
        o = oracle_lib.OracleEnv(A, T).load(inst["depot"][b], inst["task_xy"][b], inst["req"][b], inst["dur"][b])
        ref = o.rollout(int(seeds[b]), 0, oracle_lib.POLICY_RANDOM, cap_steps=20000, record=False)
        assert steps[b] == ref["n_steps"], (b, steps[b], ref["n_steps"])
        H.assert_final_matches(fin[b], ref, f"{A}A{T}T env{b}")


def test_multi_episode_rollout(gpu_device, golden_dir):
    """3 consecutive episodes, decision counter keeps running across auto-resets (manifest.json multi_episode)."""
    from dcmrta_amd.instances import generate_instance
    man = json.load(open(os.path.join(golden_dir, "manifest.json")))["multi_episode"]
    inst = generate_instance(20, 50, man["inst_seed"])
    env = _env(1, 20, 50, gpu_device)
    env.load_instances(inst["depot"][None], inst["task_xy"][None], inst["req"][None], inst["dur"][None])
    env.reset(np.array([int(man["seed_e"])], np.uint64), observe=False)
    total = 0
    for ep in man["episodes"]:
        steps = int(env.rollout_random(episodes=1).cpu()[0])
        sm = env.summary().cpu().numpy()[0]
        assert steps == ep["n_steps"]
        assert sm[0] == ep["reward"] and int(sm[1]) == ep["n_finished"]
        for i in range(6):
            assert sm[2 + i] == ep["metrics"][i]
        total += steps
    # the same three episodes in ONE launch
    env.reset(np.array([int(man["seed_e"])], np.uint64), observe=False)
    assert int(env.rollout_random(episodes=3).cpu()[0]) == total
    assert env.summary().cpu().numpy()[0, 0] == man["episodes"][-1]["reward"]


def test_hashed_traces(gpu_device):
    """G4 hashed traces: sha256 over every per-step tensor and the terminal arrays must equal the reference's."""
    from dcmrta_amd.instances import generate_instance
    hashes = H.trace_hashes()
    by_shape = {}
    for name, meta in hashes.items():
        if meta["policy"] != "random":
            continue  # nearest-policy traces need the recorded actions (only digests are committed)
        by_shape.setdefault((meta["A"], meta["T"]), []).append((name, meta))
    for (A, T), items in sorted(by_shape.items()):
        insts = [generate_instance(A, T, m["inst_seed"]) for _, m in items]
        env = _env(len(items), A, T, gpu_device)
        env.load_instances(np.stack([i["depot"] for i in insts]), np.stack([i["task_xy"] for i in insts]),
                           np.stack([i["req"] for i in insts]), np.stack([i["dur"] for i in insts]))
        seeds = np.array([int(m["seed_e"]) for _, m in items], np.uint64)
        got = H.run_lockstep(env, seeds, lambda b, i, mask, l: H.host_random_action(mask, int(items[b][1]["seed_e"]), i))
        fin = H.gpu_final(env)
        for b, (name, meta) in enumerate(items):
            assert got[b]["n_steps"] == meta["n_steps"], name
            assert float(fin[b]["reward"]) == meta["reward"], name
            assert int(fin[b]["n_finished"]) == meta["n_finished"], name


OBS_KEYS = ("leader", "action", "now", "mask", "agents_obs", "tasks_obs", "metrics", "finished", "time_start", "travel_dist",
            "agent_wait", "task_wait")


def _digest_obs(tr):
    import hashlib
    h = hashlib.sha256()
    for k in OBS_KEYS:
        h.update(np.ascontiguousarray(tr[k]).tobytes())
    return h.hexdigest()


def test_extra_hashed_traces(gpu_device, golden_dir):
    """tests/golden/trace_hashes_extra.json (reference digests over ranges-drawn sizes, max_waiting_time 3/25, loop bound
    30/250, coalition size 3, durations 2/0, more agents than tasks, coincident task locations): the sha256 over every
    per-decision output of the lockstep API (leader, action, event time, mask, both observation tensors) and the terminal
    arrays must equal the reference's; the persistent kernel must end in the same state.  The range-drawn instances run
    as ONE ragged batch."""
    from test_oracle_golden import extra_instance
    hashes = json.load(open(os.path.join(golden_dir, "trace_hashes_extra.json")))
    groups = {}
    for name, meta in hashes.items():
        if meta["policy"] != "random":
            continue  # nearest-policy traces need the recorded actions (only digests are committed)
        key = ("ranges",) if meta["kind"] == "ranges" else (meta["A"], meta["T"], float(meta.get("max_waiting_time", 10.0)),
                                                           float(meta.get("max_time", 100.0)))
        groups.setdefault(key, []).append((name, meta))
    assert len(groups) >= 10
    for key, items in groups.items():
        insts = [extra_instance(m) for _, m in items]
        B = len(items)
        nA = np.array([a for a, _ in insts], np.int32)
        nT = np.array([len(i["req"]) for _, i in insts], np.int32)
        if key == ("ranges",):
            A, T, mwt, mt = 20, 50, 10.0, 100.0
            kw = dict(n_agents=nA, n_tasks=nT)
        else:
            (A, T, mwt, mt), kw = key, {}
        pad = lambda x, n: np.concatenate([x, np.ones((n - len(x),) + x.shape[1:], x.dtype)])
        env = _env(B, A, T, gpu_device, max_waiting_time=mwt, max_time=mt)
        env.load_instances(np.stack([i["depot"] for _, i in insts]), np.stack([pad(i["task_xy"], T) for _, i in insts]),
                           np.stack([pad(i["req"], T) for _, i in insts]), np.stack([pad(i["dur"], T) for _, i in insts]), **kw)
        seeds = np.array([int(m["seed_e"]) for _, m in items], np.uint64)
        got = H.run_lockstep(env, seeds, lambda b, i, mask, l: H.host_random_action(mask[:nT[b] + 1], int(seeds[b]), i))
        fin = H.gpu_final(env)
        env.reset(seeds, observe=False)
        ksteps = env.rollout_random(1).cpu().numpy()
        kfin = H.gpu_final(env)
        for b, (name, meta) in enumerate(items):
            a, t = int(nA[b]), int(nT[b])
            g, f = got[b], fin[b]
            assert g["n_steps"] == meta["n_steps"] == ksteps[b], name
            tr = dict(leader=g["leader"], action=g["action"], now=g["now"], mask=g["mask"][:, :t + 1],
                      agents_obs=g["agents_obs"][:, :a], tasks_obs=g["tasks_obs"][:, :t + 1], metrics=f["metrics"],
                      finished=f["finished"][:t].astype(np.uint8), time_start=f["time_start"][:t], travel_dist=f["travel_dist"][:a],
                      agent_wait=f["agent_wait"][:a], task_wait=f["task_wait"][:t])
            assert _digest_obs(tr) == meta["sha256_obs"], name
            assert float(f["reward"]) == meta["reward"] == float(kfin[b]["reward"]), name
            assert int(f["n_finished"]) == meta["n_finished"] == int(kfin[b]["n_finished"]), name
            for k in ("time_start", "travel_dist", "agent_wait", "task_wait", "metrics"):
                assert np.array_equal(f[k], kfin[b][k], equal_nan=True), (name, k)
