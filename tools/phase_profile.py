#!/usr/bin/env python3
"""Per-phase wave-cycle breakdown of k_rollout_random / k_step (developer tool, run on the GPU box).

Builds a SEPARATE library (tools/libdcmrta_prof.so, -DDCM_PROFILE_PHASES) whose kernels accumulate
s_memtime deltas per phase, runs the config-2 workload and prints cycles per decision and share per phase.

    python tools/phase_profile.py [B A T]                 persistent kernel
    python tools/phase_profile.py lockstep B A T [N]      k_step: mean and max over envs of every phase, N batched steps"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.path.join(ROOT, "tools", "libdcmrta_prof.so")
src = os.path.join(ROOT, "dcmrta_amd", "csrc")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-mllvm", "-phi-elim-split-all-critical-edges=1",
                       "-fPIC", "-shared", "-DDCM_PROFILE_PHASES", os.path.join(src, "dcmrta_env.hip"),
                       os.path.join(src, "dcmrta_replay.hip"), "-o", so])
import torch  # noqa: E402
from dcmrta_amd import _lib  # noqa: E402
_lib.LIB_PATH = so
from dcmrta_amd.batched_env import BatchedTaskEnv  # noqa: E402
from dcmrta_amd.choice import env_seeds  # noqa: E402
from dcmrta_amd.instances import generate_batch  # noqa: E402

if len(sys.argv) > 1 and sys.argv[1] == "lockstep":
    B, A, T = (int(x) for x in (sys.argv[2:5] if len(sys.argv) > 4 else (4096, 20, 50)))
    N = int(sys.argv[5]) if len(sys.argv) > 5 else 60
    env = BatchedTaskEnv(B, A, T).load_instances(**generate_batch(B, A, T, 0))
    obs = env.reset(env_seeds(0, 0, B))
    import numpy as np
    raw = C.CDLL(so)
    rows = np.zeros((B, 32), np.uint64)
    for i in range(10):                                         # warm-up
        obs = env.step(torch.multinomial((~obs.mask).float(), 1).squeeze(1).int())
    torch.cuda.synchronize()
    raw.dcm_prof_read_step(rows.ctypes.data_as(C.c_void_p), B, 1)
    ev = []
    for i in range(N):
        act = torch.multinomial((~obs.mask).float(), 1).squeeze(1).int()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); obs = env.step(act); e1.record()
        ev.append((e0, e1))
    torch.cuda.synchronize()
    raw.dcm_prof_read_step(rows.ctypes.data_as(C.c_void_p), B, 0)
    rows = rows.astype(np.float64)
    waves = float(B) * N
    act_waves = rows[:, 28].sum()
    ms = sorted(a.elapsed_time(b) for a, b in ev)[N // 2]
    print(f"k_step {B} x {A}A/{T}T, {N} batched steps; HIP-event median {ms * 1e3:.1f} us per batched step (instrumented build); "
          f"active env-steps {rows[:, 2].astype(bool).sum()} envs")
    names = ["record HBM->LDS", "key + leader", "apply+updates+advance", "auto-reset", "next leader + observe", "write-back issue",
             "whole wave"]
    for i, nm in enumerate(names):
        print(f"  {nm:24s} mean {rows[:, i].sum() / waves:9.0f} cyc   max over all waves {rows[:, 8 + i].max():9.0f} cyc   "
              f"mean of per-env maxima {rows[:, 8 + i].mean():9.0f}")
    inner = ["", "", "", "apply: move+slots", "task_update#1", "agent_update#1", "advance:D+groups", "task_update#2", "agent_update#2",
             "advance:tail(+terminal)", "", "apply: followers+target"]
    for i, nm in enumerate(inner):
        if nm:
            print(f"    {nm:24s} mean {rows[:, 16 + i].sum() / waves:9.1f} cyc per wave")
    sys.exit(0)
B, A, T = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (4096, 20, 50)))
env = BatchedTaskEnv(B, A, T).load_instances(**generate_batch(B, A, T, 0))
env.reset(env_seeds(0, 0, B), observe=False)
env.rollout_random(1)
raw = C.CDLL(so)
buf = (C.c_ulonglong * 16)()
torch.cuda.synchronize()
raw.dcm_prof_read(buf, 1)
n = int(env.rollout_random(1).sum())
torch.cuda.synchronize()
raw.dcm_prof_read(buf, 0)
names = ["leader", "observe", "action", "apply: move+slots", "task_update#1", "agent_update#1", "advance:D+groups",
         "task_update#2", "agent_update#2", "advance:tail", "reset+first event", "apply: followers+target"]
tot = sum(buf[i] for i in range(12))
print(f"{n} decisions, {tot / n:.0f} wave-cycles (s_memtime) per decision")
for i, nm in enumerate(names):
    print(f"  {nm:22s} {buf[i] / n:8.1f} cyc/decision  {100.0 * buf[i] / max(tot, 1):5.1f} %")
for i, nm in ((12, "terminal: task waits"), (13, "terminal: agent waits"), (14, "terminal: pairwise sums"), (15, "terminal: whole call")):
    print(f"  {nm:22s} {buf[i] / n:8.1f} cyc/decision  (inside advance:tail)")
