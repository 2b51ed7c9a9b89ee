"""Child script of tests/test_launch.py: the shape of bench.py's start-up (self-launch first, then one rank per process over a
process group) without the HIP env, so that the launch path is covered on a CPU-only machine."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    from dcmrta_amd.launch import maybe_self_launch
    assert "torch" not in sys.modules                      # the launcher decision is taken before torch is imported
    maybe_self_launch(__file__)

    import argparse
    import torch
    from dcmrta_amd.dist import DistContext, shard_range
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--fail-rank", type=int, default=-1)
    ap.add_argument("--envs", type=int, default=13)
    ap.add_argument("--hang-rank", type=int, default=-1, help="this rank sleeps forever BEFORE joining the process group")
    ap.add_argument("--dist-timeout", type=float, default=120.0)
    ap.add_argument("--launch-timeout", type=float, default=None)
    ap.add_argument("--preflight", action="store_true", help="let DistContext pick the device itself (device-count preflight)")
    a = ap.parse_args()
    if int(os.environ.get("RANK", "0")) == a.hang_rank:
        import time
        time.sleep(10 ** 6)
    if a.preflight:
        ctx = DistContext.from_env(expected_world=a.gpus, backend="gloo")
    else:
        ctx = DistContext.from_env(expected_world=a.gpus, backend="gloo", device=torch.device("cpu"), timeout_s=a.dist_timeout)
    if ctx.rank == a.fail_rank:
        sys.exit(7)
    lo, hi = shard_range(a.envs, ctx.rank, ctx.world)
    local = torch.arange(lo, hi, dtype=torch.float64)
    g = ctx.all_gather_returns(local, n_total=a.envs)
    ctx.verify_gather(g, local, lo)
    total = ctx.sum_over_ranks(hi - lo)
    names = ctx.device_names()                           # (a collective: every rank calls it)
    if ctx.rank == 0:
        print(json.dumps({"n_gpus": ctx.world, "process_group_ranks": ctx.group_size(), "envs": total,
                          "gathered": g.tolist(), "rank_devices": names, "self_launched": os.environ.get("DCM_SELF_LAUNCHED")}), flush=True)
    ctx.shutdown()
