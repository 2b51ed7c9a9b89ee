"""-m gpu: the C ABI is self-sufficient -- examples/c_abi_demo.cpp (plain C++/HIP, no Python, no torch) is compiled against
include/dcmrta_env.h + libdcmrta_hip.so, run, and its episode statistics are checked against the Python path."""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_demo_runs_and_matches_python(gpu_device, tmp_path):
    exe = str(tmp_path / "c_abi_demo")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", os.path.join(ROOT, "examples", "c_abi_demo.cpp"),
                           "-I" + os.path.join(ROOT, "include"), "-L" + os.path.join(ROOT, "dcmrta_amd"), "-ldcmrta_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "dcmrta_amd"), "-o", exe])
    out = json.loads(subprocess.check_output([exe, "64"]).decode().strip().splitlines()[-1])
    assert out["envs"] == 64 and out["decisions"] > 64 * 50
    # same instances / seeds through the Python binding
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.choice import env_seeds
    B, A, T = 64, 20, 50
    s = 12345
    vals = []
    for _ in range(2 * B + 2 * B * T + B * T):
        s = (s * 6364136223846793005 + 1442695040888963407) % (1 << 64)
        vals.append((s >> 11) / 9007199254740992.0)
    v = np.array(vals)
    depot, xy = v[:2 * B].reshape(B, 2), v[2 * B:2 * B + 2 * B * T].reshape(B, T, 2)
    req = (1 + (v[2 * B + 2 * B * T:] * 5.0).astype(np.int32)).reshape(B, T)
    env = BatchedTaskEnv(B, A, T, device=gpu_device).load_instances(depot, xy, req, np.full((B, T), 5.0))
    env.reset(env_seeds(0, 0, B), observe=False)
    steps = env.rollout_random(1, write_obs=False)
    sm = env.summary().cpu().numpy()
    assert int(steps.sum()) == out["decisions"]
    assert abs(sm[:, 3].mean() - out["mean_makespan"]) < 1e-6
