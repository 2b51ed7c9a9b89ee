#!/usr/bin/env python3
"""Per-phase wave-cycle breakdown of k_rollout_random (developer tool, run on the GPU box).

Builds a SEPARATE library (tools/libdcmrta_prof.so, -DDCM_PROFILE_PHASES) whose rollout kernel accumulates
s_memtime deltas per phase, runs the config-2 workload and prints cycles per decision and share per phase."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.path.join(ROOT, "tools", "libdcmrta_prof.so")
src = os.path.join(ROOT, "dcmrta_amd", "csrc")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                       "-fPIC", "-shared", "-DDCM_PROFILE_PHASES", os.path.join(src, "dcmrta_env.hip"),
                       os.path.join(src, "dcmrta_replay.hip"), "-o", so])
import torch  # noqa: E402
from dcmrta_amd import _lib  # noqa: E402
_lib.LIB_PATH = so
from dcmrta_amd.batched_env import BatchedTaskEnv  # noqa: E402
from dcmrta_amd.choice import env_seeds  # noqa: E402
from dcmrta_amd.instances import generate_batch  # noqa: E402

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
