"""-m gpu: route replay (execute_by_route, env/task_env.py:562-593) on the device.

 * the reference-published known answer: CTAS-D routes -> metrics/metrics.csv:2 (via the committed G3 outputs)
 * dynamic task visibility (reactive_planning) incl. the 4 instances where the reference raises TypeError
 * BASELINE config-5 size (100A/500T) against the oracle on synthetic routes"""
import os

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu
KEYS_EXACT = ("finished", "time_start", "time_finish", "task_wait", "n_members", "travel_dist", "returned", "agent_wait")


def _check(out, b, ref, name):
    for k in KEYS_EXACT:
        got = out[k][b].cpu().numpy()
        assert np.array_equal(got.astype(np.asarray(ref[k]).dtype), ref[k]), (name, k)
    sm = out["summary"][b].cpu().numpy()
    m = ref["metrics"]
    assert sm[3] == ref["makespan"] and sm[0] == -ref["makespan"], name
    for i in range(6):
        assert sm[2 + i] == m[i], (name, i)


@pytest.mark.parametrize("reactive,fixture", [(False, "ctasd_replay"), (True, "reactive_replay")])
def test_ctasd_routes_known_answer(gpu_device, golden_dir, reactive, fixture):
    from dcmrta_amd import _lib
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.instances import load_instances_npz, load_routes_json
    inst, A = load_instances_npz(os.path.join(golden_dir, "instances_20A50T.npz"))
    routes = load_routes_json(os.path.join(golden_dir, "ctasd_routes.json"))
    g = np.load(os.path.join(golden_dir, fixture + ".npz"))
    env = BatchedTaskEnv(50, A, 50, device=gpu_device)
    env.load_instances(**inst)
    # baselines/CTAS-D.py:41-45: a vehicle whose node list is [0] keeps pre_set_route None; otherwise nodes[1:]
    rl = [[(None if r == [0] else r[1:]) for r in routes[i]] + [None] * (A - len(routes[i])) for i in range(50)]
    env.load_routes(rl)
    out = env.execute_routes(reactive=reactive)
    flags = out["flags"].cpu().numpy()
    idx, raised = list(g["idx"]), list(g["raised"])
    for i in range(50):
        if i in raised:
            assert flags[i] & _lib.FLAG_TYPE_ERROR, i      # env/task_env.py:220 raises in the reference
            continue
        assert not (flags[i] & (_lib.FLAG_TYPE_ERROR | _lib.FLAG_OVERFLOW | _lib.FLAG_TRUNCATED | _lib.FLAG_BAD_ACTION)), i
        k = idx.index(i)
        _check(out, i, {key: g[key][k] for key in g.files if key not in ("idx", "raised")}, f"{fixture}[{i}]")
    if not reactive:
        sm = out["summary"].cpu().numpy()
        # metrics/metrics.csv:2 of the reference: CTAS-D_300s 36.908 makespan, 5.619 waiting, 42.027 travel, 2.248 efficiency
        assert round(sm[:, 3].mean(), 3) == 36.908 and round(sm[:, 5].mean(), 3) == 5.619
        assert round(sm[:, 6].mean(), 3) == 42.027 and round(sm[:, 7].mean(), 3) == 2.248


from dcmrta_amd.instances import synthetic_routes  # noqa: E402


@pytest.mark.parametrize("A,T,reactive,cap", [(100, 500, False, 8), (100, 500, True, 8), (100, 100, True, 8), (50, 200, True, 8),
                                              (13, 37, False, 8),
                                              # bench.py --config 5's setting: 5 member slots
                                              (100, 500, False, 5), (100, 500, True, 5)])
# "auto": the register-resident kernel (replay_fast.hpp) when the live tasks fit two lane chunks -- 13A/37T, and every reactive case
# at the reference's visibility cap of 100 -- else the general kernel with its own choice; "lds" / "hbm": the general kernel with
# its replay scratch there
@pytest.mark.parametrize("placement", ["auto", "lds", "hbm"])
def test_replay_matches_oracle(gpu_device, oracle_lib, A, T, reactive, cap, placement):
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.instances import generate_batch
    B = 3
    inst = generate_batch(B, A, T, base_seed=900 + T)
    rl = [synthetic_routes(inst["req"][b], A, max_task=100 if reactive else None) for b in range(B)]
    env = BatchedTaskEnv(B, A, T, device=gpu_device)
    env.load_instances(**inst)
    env.load_routes(rl, member_cap=cap)
    env.set_replay_placement(placement)
    out = env.execute_routes(reactive=reactive)
    flags = out["flags"].cpu().numpy()
    assert not (flags & 0x78).any()
    for b in range(B):
        o = oracle_lib.OracleEnv(A, T).load(inst["depot"][b], inst["task_xy"][b], inst["req"][b], inst["dur"][b])
        for a, r in enumerate(rl[b]):
            o.pre_set_route(r, a)
        ref = o.execute_by_route(reactive)
        # With more than 100 tasks the reference's reactive mode never terminates (visibility is hard-capped at 100,
        # env/task_env.py:567): the zero-decider guard shared by oracle and kernel ends such episodes identically.
        assert bool(flags[b] & 4) == bool(ref["truncated"]) == (reactive and T > 100)
        _check(out, b, ref, f"{A}A{T}T reactive={reactive} env{b}")


def test_random_route_replays_known_answers(gpu_device, golden_dir):
    """tests/golden/replay_random.json: the reference's own execute_by_route results on random routes (surplus visitors
    released before they arrive -> non-monotone arrival lists, too few visitors, None routes, shuffled order)."""
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from test_oracle_golden import random_replay_cases
    n_ok = 0
    for c, inst in random_replay_cases(golden_dir):
        env = BatchedTaskEnv(1, c["A"], c["T"], device=gpu_device)
        env.load_instances(inst["depot"][None], inst["task_xy"][None], inst["req"][None], inst["dur"][None])
        env.load_routes([c["routes"]], member_cap=16)
        out = env.execute_routes(reactive=c["reactive"])
        flags = int(out["flags"].cpu().numpy()[0])
        name = (c["A"], c["T"], c["reactive"])
        if c["status"] == "type_error":
            assert flags & 64, name
        elif c["status"] == "no_termination":
            assert flags & 4 and not flags & 64, name
        else:
            assert not flags & (4 | 16 | 64), name
            for k in KEYS_EXACT:
                exp = np.asarray(c["result"][k])
                assert np.array_equal(out[k][0].cpu().numpy().astype(exp.dtype), exp), (name, k)
            sm = out["summary"][0].cpu().numpy()
            assert np.array_equal(sm[2:8], np.asarray(c["result"]["metrics"])), name
            n_ok += 1
        env.close()
    assert n_ok >= 15


@pytest.mark.parametrize("cap", [8, 5])
def test_random_route_replays_on_the_register_resident_kernel(gpu_device, golden_dir, cap):
    """The same reference known answers through replay_fast.hpp (member_cap <= 8 and the default placement select it for these
    shapes); cases in which a task collects more members than the slots hold are flagged DCM_FLAG_OVERFLOW and not compared."""
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from test_oracle_golden import random_replay_cases
    n_ok = n_over = 0
    for c, inst in random_replay_cases(golden_dir):
        env = BatchedTaskEnv(1, c["A"], c["T"], device=gpu_device)
        env.load_instances(inst["depot"][None], inst["task_xy"][None], inst["req"][None], inst["dur"][None])
        env.load_routes([c["routes"]], member_cap=cap)
        out = env.execute_routes(reactive=c["reactive"])
        flags = int(out["flags"].cpu().numpy()[0])
        name = (c["A"], c["T"], c["reactive"], cap)
        if flags & 16:
            n_over += 1
        elif c["status"] == "type_error":
            assert flags & 64, name
        elif c["status"] == "no_termination":
            assert flags & 4 and not flags & 64, name
        else:
            assert not flags & (4 | 64), name
            for k in KEYS_EXACT:
                exp = np.asarray(c["result"][k])
                assert np.array_equal(out[k][0].cpu().numpy().astype(exp.dtype), exp), (name, k)
            assert np.array_equal(out["summary"][0].cpu().numpy()[2:8], np.asarray(c["result"]["metrics"])), name
            n_ok += 1
        env.close()
    assert n_ok >= (10 if cap == 8 else 5), (n_ok, n_over)


def test_generalised_visibility_schedule(gpu_device, golden_dir, oracle_lib):
    """dcm_set_visibility: the reactive replay under other dynamic-arrival constants than the reference's 20 / 20 / 10 / 100,
    against the reference run with those constants (tests/golden/replay_schedule.json) and, at BASELINE config-5 size with a
    schedule under which all 500 tasks appear, against the oracle on a small batch."""
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.instances import generate_batch
    from test_oracle_golden import schedule_cases
    n_ok = 0
    for c, inst in schedule_cases(golden_dir):
        env = BatchedTaskEnv(1, c["A"], c["T"], device=gpu_device)
        env.load_instances(inst["depot"][None], inst["task_xy"][None], inst["req"][None], inst["dur"][None])
        env.set_visibility(*c["schedule"])
        env.load_routes([c["routes"]], member_cap=16)
        out = env.execute_routes(reactive=True)
        flags = int(out["flags"].cpu().numpy()[0])
        name = (c["A"], c["T"], tuple(c["schedule"]))
        if c["status"] == "type_error":
            assert flags & 64, name
        elif c["status"] == "no_termination":
            assert flags & 4 and not flags & 64, name
        else:
            assert not flags & (4 | 16 | 64), name
            for k in KEYS_EXACT:
                exp = np.asarray(c["result"][k])
                assert np.array_equal(out[k][0].cpu().numpy().astype(exp.dtype), exp), (name, k)
            assert np.array_equal(out["summary"][0].cpu().numpy()[2:8], np.asarray(c["result"]["metrics"])), name
            n_ok += 1
        env.close()
    assert n_ok >= 6
    # config-5 size, every task routed, all of them eventually visible
    B, A, T = 3, 100, 500
    inst = generate_batch(B, A, T, base_seed=77)
    rl = [synthetic_routes(inst["req"][b], A) for b in range(B)]
    env = BatchedTaskEnv(B, A, T, device=gpu_device)
    env.load_instances(**inst)
    env.set_visibility(100, 100, 10, 500)
    env.load_routes(rl, member_cap=8)
    out = env.execute_routes(reactive=True)
    assert not (out["flags"].cpu().numpy() & 0x78).any()
    for b in range(B):
        o = oracle_lib.OracleEnv(A, T).load(inst["depot"][b], inst["task_xy"][b], inst["req"][b], inst["dur"][b])
        o.set_visibility(100, 100, 10, 500)
        for a, r in enumerate(rl[b]):
            o.pre_set_route(r, a)
        _check(out, b, o.execute_by_route(True), f"all-visible schedule env{b}")
    with pytest.raises(Exception):
        env.set_visibility(20, 0, 10, 100)


def test_replay_placements_agree_at_config5_size(gpu_device):
    """Full-size property (BASELINE config 5 shape, a batch large enough for the HBM placement to be the automatic one): the replay
    gives bit-identical results whether its scratch block lives in LDS or in HBM, and the same again on a second run."""
    import torch
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.instances import generate_batch, synthetic_route_arrays
    B, A, T = 2048, 100, 500
    inst = generate_batch(B, A, T, base_seed=5, first=0)
    routes, route_len = synthetic_route_arrays(inst["req"], A, max_task=100)
    outs = {}
    for pl in ("lds", "hbm", "auto", "hbm2"):
        env = BatchedTaskEnv(B, A, T, device=gpu_device)
        env.load_instances(**inst)
        env.load_route_arrays(routes, route_len, member_cap=5)
        env.set_replay_placement(pl.rstrip("2"))
        o = env.execute_routes(True)
        outs[pl] = {k: v.clone() for k, v in o.items() if isinstance(v, torch.Tensor)}
        env.close()
    ref = outs["lds"]
    assert int(ref["steps"].sum()) > 100 * B and not (ref["flags"] & 0x78).any()
    for pl in ("hbm", "auto", "hbm2"):
        for k, v in ref.items():
            assert torch.equal(v.view(torch.uint8) if v.dtype.is_floating_point else v,
                               outs[pl][k].view(torch.uint8) if v.dtype.is_floating_point else outs[pl][k]), (pl, k)
