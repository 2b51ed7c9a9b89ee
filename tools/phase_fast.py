#!/usr/bin/env python3
"""Per-phase wave-cycle breakdown of the register-resident persistent kernel k_rollout_fast (developer tool, GPU box).

    python tools/phase_fast.py [A T episodes] [B ...]       (default 20 50 3, B = 4096 1024 256)

Separate -DDCM_PROFILE_PHASES build (tools/_variants/lib_prof.so, built if absent): s_memtime marks inside Fast<>::decide / apply /
next_event accumulate per phase; every mark costs a scalar-memory round trip that lands in the phase it closes, so read the shares,
not the absolute clocks.  B = 4096 is four waves per SIMD (the BASELINE batch), 1024 one wave per SIMD (a wave's own latency)."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SRC = os.path.join(ROOT, "dcmrta_amd", "csrc")
SO = os.path.join(ROOT, "tools", "_variants", "lib_prof.so")
NAMES = ["key + leader", "observe", "action", "vacancy + followers", "target + distance", "join + member update", "task_update #1",
         "agent_update #1", "group bookkeeping", "next_event: nanmin + groups", "task_update #2", "agent_update #2", "loop latch",
         "general path (advance / reset / reload / flush)"]


def main():
    args = [int(x) for x in sys.argv[1:]]
    A, T, EP = (args[:3] + [20, 50, 3][len(args[:3]):])
    Bs = args[3:] or [4096, 1024, 256]
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    if not os.path.exists(SO):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-mllvm", "-phi-elim-split-all-critical-edges=1",
                               "-fPIC", "-shared", "-DDCM_PROFILE_PHASES", os.path.join(SRC, "dcmrta_env.hip"),
                               os.path.join(SRC, "dcmrta_replay.hip"), "-o", SO])
    os.environ["DCMRTA_HIP_LIB"] = SO
    import torch
    from dcmrta_amd import _lib
    from dcmrta_amd.batched_env import BatchedTaskEnv
    from dcmrta_amd.choice import env_seeds
    from dcmrta_amd.instances import generate_batch
    lib = _lib.load()
    lib.dcm_prof_read_fast.restype = C.c_int
    lib.dcm_prof_read_fast.argtypes = [C.c_void_p, C.c_int]
    buf = (C.c_ulonglong * 16)()
    for B in Bs:
        env = BatchedTaskEnv(B, A, T, device="cuda:0")
        env.load_instances(**generate_batch(B, A, T, base_seed=0))
        env.reset(env_seeds(0, 0, B), observe=False)
        env.rollout_random(episodes=EP)                                  # warm
        torch.cuda.synchronize()
        _lib.check(lib.dcm_prof_read_fast(buf, 1))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        steps = env.rollout_random(episodes=EP)
        e1.record()
        torch.cuda.synchronize()
        _lib.check(lib.dcm_prof_read_fast(buf, 0))
        n = int(steps.sum().item())
        mx = int(steps.max().item())
        tot = sum(buf[i] for i in range(16))
        print(f"B {B} x {A}A/{T}T x {EP} episodes: {n} decisions (max per env {mx}), launch {e0.elapsed_time(e1) * 1e3:.0f} us (instrumented), "
              f"{tot / n:.0f} marked clocks per decision")
        for i, nm in enumerate(NAMES):
            print(f"   {nm:48s} {buf[i] / n:8.1f}  {100.0 * buf[i] / tot:5.1f} %")
        env.close()


if __name__ == "__main__":
    main()
