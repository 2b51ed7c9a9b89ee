#!/usr/bin/env python3
"""Self-check of the incremental task_update (developer tool, GPU box): builds the library with -DDCM_INC_DEBUG, whose
kernels dry-run the skipped tasks of every incremental call and printf a line whenever a full pass would have changed one
of them, then rolls out a few multi-chunk shapes.  Expected output: only the `checked` lines."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.path.join(ROOT, "tools", "libdcmrta_dbg.so")
src = os.path.join(ROOT, "dcmrta_amd", "csrc")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-mllvm", "-phi-elim-split-all-critical-edges=1",
                       "-shared", "-DDCM_INC_DEBUG", os.path.join(src, "dcmrta_env.hip"), os.path.join(src, "dcmrta_replay.hip"), "-o", so])
from dcmrta_amd import _lib  # noqa: E402
_lib.LIB_PATH = so
import torch  # noqa: E402
from dcmrta_amd.batched_env import BatchedTaskEnv  # noqa: E402
from dcmrta_amd.choice import env_seeds  # noqa: E402
from dcmrta_amd.instances import generate_batch  # noqa: E402

for (A, T, mwt, mt) in ((50, 200, 10.0, 100.0), (32, 200, 3.0, 100.0), (100, 500, 10.0, 100.0), (13, 130, 3.0, 250.0), (70, 65, 25.0, 100.0)):
    B = 24
    env = BatchedTaskEnv(B, A, T, max_waiting_time=mwt, max_time=mt).load_instances(**generate_batch(B, A, T, base_seed=A * T))
    env.reset(env_seeds(A + T, 0, B), observe=False)
    n = int(env.rollout_random(2).sum())
    torch.cuda.synchronize()
    print(f"checked {A}A/{T}T mwt={mwt} max_time={mt}: {n} decisions", flush=True)
