"""CPU: the oracle under AddressSanitizer + UBSan (sanitizers are CPU-only on this pool).

oracle/Makefile's liboracle_asan.so target is built and, in a child process with the sanitizer runtime preloaded, replays three
golden traces bit for bit, a route replay, and the multi-threaded batch runners (batch_rollout_full / batch_replay: the functions
behind bench.py's parity and cpu_baseline legs).  Any report makes the child exit non-zero (halt_on_error)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import glob, os, sys
import numpy as np
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import helpers as H
import oracle
assert "asan" in os.environ["DCM_ORACLE_LIB"]
POLICY = {"random": 0, "first": 2, "nearest": 3, "anymask": 4}
paths = [p for p in H.full_traces() if any(s in p for s in ("trace_20A50T_random_s0", "trace_5A8T_anymask_s1", "trace_50A200T_nearest_s0"))]
assert len(paths) == 3, paths
for p in paths:
    tr = H.load_trace(p)
    e = oracle.OracleEnv(int(tr["A"]), int(tr["T"])).load(tr["depot"], tr["task_xy"], tr["req"], tr["dur"])
    out = e.rollout(int(tr["seed_e"]), 0, POLICY[H.trace_policy(p)], cap_steps=4096)
    assert out["n_steps"] == int(tr["n_steps"]) and out["reward"] == float(tr["reward"])
    for k in ("leader", "action", "now", "mask", "agents_obs", "tasks_obs", "metrics", "time_start", "agent_wait"):
        assert np.array_equal(np.asarray(out[k]), tr[k]), (p, k)
from dcmrta_amd.choice import env_seeds
from dcmrta_amd.instances import generate_batch, synthetic_route_arrays
B, A, T = 48, 20, 50
inst = generate_batch(B, A, T, base_seed=1)
seeds = env_seeds(1, 0, B)
r4 = oracle.batch_rollout_full(inst["depot"], inst["task_xy"], inst["req"], inst["dur"], seeds, A, episodes=2, threads=4)
r1 = oracle.batch_rollout_full(inst["depot"], inst["task_xy"], inst["req"], inst["dur"], seeds, A, episodes=2, threads=1)
assert r4["total"] == r1["total"] > 0 and all(np.array_equal(r4[k], r1[k]) for k in ("steps", "returns", "metrics", "n_finished"))
B, A, T = 6, 100, 500
inst = generate_batch(B, A, T, base_seed=2)
routes, rl = synthetic_route_arrays(inst["req"], A, max_task=100)
p4 = oracle.batch_replay(inst["depot"], inst["task_xy"], inst["req"], inst["dur"], routes, rl, True, (20, 20, 10, 100), threads=3)
o = oracle.OracleEnv(A, T).load(inst["depot"][2], inst["task_xy"][2], inst["req"][2], inst["dur"][2])
for a in range(A):
    if rl[2, a] >= 0:
        o.pre_set_route(routes[2, a, :rl[2, a]], a)
ref = o.execute_by_route(True)
assert ref["reward"] == p4["reward"][2] and int(ref["route_len"].sum()) == p4["steps"][2] and np.array_equal(ref["metrics"], p4["metrics"][2])
print("sanitized oracle ok")
"""


def test_oracle_is_clean_under_asan_and_ubsan():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle_asan.so"], check=True, timeout=300)
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True, check=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan runtime next to gcc")
    env = dict(os.environ, LD_PRELOAD=asan, DCM_ORACLE_LIB=os.path.join(ROOT, "oracle", "liboracle_asan.so"),
               # python itself leaks by design; everything else is fatal
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0:exitcode=66", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    out = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + CHILD], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0 and "sanitized oracle ok" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]
    assert "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr, out.stderr[-3000:]
