"""-m gpu: route history (agent['route'] / agent['arrival_time'], env/task_env.py:95-96,314,318) recorded by the lockstep API
equals the oracle's lists; export helpers of dcmrta_amd/trajectory.py."""
import os

import numpy as np
import pytest
import yaml

import helpers as H

pytestmark = pytest.mark.gpu


def test_route_log_equals_oracle_routes(gpu_device, oracle_lib, golden_dir, tmp_path):
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.trajectory import route_history, routes_to_yaml, write_results_csv
    names = ["trace_20A50T_random_s0", "trace_20A50T_nearest_s1", "trace_20A50T_first_s0"]
    traces = [H.load_trace(os.path.join(golden_dir, n + ".npz")) for n in names]
    env = BatchedTaskEnv(len(traces), 20, 50, device=gpu_device).enable_route_log(cap=32)
    env.load_instances(np.stack([t["depot"] for t in traces]), np.stack([t["task_xy"] for t in traces]),
                       np.stack([t["req"] for t in traces]), np.stack([t["dur"] for t in traces]))
    seeds = np.array([int(t["seed_e"]) for t in traces], np.uint64)
    H.run_lockstep(env, seeds, lambda b, i, m, l: int(traces[b]["action"][i]))
    for b, tr in enumerate(traces):
        o = oracle_lib.OracleEnv(20, 50).load(tr["depot"], tr["task_xy"], tr["req"], tr["dur"])
        o.rollout(int(tr["seed_e"]), 0, oracle_lib.POLICY_INJECTED, inj_action=tr["action"], record=False)
        hist = route_history(env, b)
        assert [len(r) for r, _ in hist] == tr["route_len"].tolist()           # reference's own route lengths
        for a in range(20):
            rt, ra = o.route(a)
            assert hist[a][0] == rt.tolist() and hist[a][1] == ra.tolist()
    routes = routes_to_yaml(env, tmp_path / "route.yaml", b=0)                  # worker.py:244-251 numbering: 0 = depot
    back = yaml.safe_load(open(tmp_path / "route.yaml"))
    assert back == routes and all(min(r) >= 0 for r in routes.values() if r)
    write_results_csv(tmp_path / "res.csv", env.summary())
    lines = open(tmp_path / "res.csv").read().splitlines()
    assert lines[0].split(",")[1:] == ["success_rate", "makespan", "time_cost", "waiting_time", "travel_dist", "efficiency"]
    assert float(lines[1].split(",")[2]) == float(traces[0]["makespan"])
    # a second reset clears the log
    env.reset(seeds, observe=False)
    assert int(env.routes()[2].sum()) == 0


@pytest.mark.parametrize("name", ["traj_5A8T_random_s3.npz", "traj_10A20T_nearest_s4.npz", "traj_6A9T_random_s5.npz"])
def test_generate_traj_from_device_state(gpu_device, golden_dir, name):
    """generate_traj (env/task_env.py:375-418) from the device state alone -- route log, dcm_get_members, task times --
    equals the trajectories the reference produced for the same episode (recorded actions, protocol leaders/followers)."""
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.trajectory import generate_traj
    from test_host import _traj_fixture
    z, routes, members, ref = _traj_fixture(golden_dir, name)
    A, T = z["route"].shape[0], z["task_xy"].shape[0]
    env = BatchedTaskEnv(1, A, T, device=gpu_device).enable_route_log(cap=64)
    env.load_instances(z["depot"][None], z["task_xy"][None], z["req"][None], z["dur"][None])
    H.run_lockstep(env, np.array([int(z["seed_e"])], np.uint64), lambda b, i, m, l: int(z["action"][i]))
    mem = env.task_members()[0].cpu().numpy()
    assert np.array_equal(mem, z["members"])                                   # final task['members'] lists, in order
    got = generate_traj(env, 0)
    for a, (g, r) in enumerate(zip(got, ref)):
        assert g.shape == r.shape and np.array_equal(g, r), (name, a)
