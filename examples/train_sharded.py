#!/usr/bin/env python3
"""Data-parallel REINFORCE rounds with one process per GPU (the sharded form of driver.py:99-199).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29600 \
        examples/train_sharded.py --total-envs 4096 --rounds 4

Every rank owns one GPU and one BatchedRunner with its shard of the round's env budget (dcmrta_amd/dist_runner.py).  Per round:
(agents_num, tasks_num) are drawn on rank 0 and broadcast (driver.py:114-115), every rank plays its shard (sampled episode +
greedy twin), ONE all-gather moves the per-env terminal rows, the REINFORCE loss (driver.py:175-188) is computed on the
rank-local experience and the gradients are all-reduced, so all ranks take the same Adam step and the weights never have to be
sent again.  `--learner rank0` instead gathers the experience to rank 0 (the reference's single learner) and re-broadcasts the
weights.  With one rank it is a plain single-GPU training loop.  On a 1-GPU box: DCM_FORCE_DEVICE=0 DCM_DIST_BACKEND=gloo.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dcmrta_amd.dist import DistContext  # noqa: E402
from dcmrta_amd.dist_runner import ShardedRunner  # noqa: E402
from dcmrta_amd.policy import AttentionNet  # noqa: E402


def reinforce_backward(net, res, scale=1.0, minibatch=8192):
    """driver.py:175-182 on a set of decisions: back-propagates scale * sum_i -logp_i(action_i) * advantage_i.  The decisions are
    fed through the net `minibatch` at a time and the gradients accumulate, which is the same sum without holding the activations
    of a whole round (5e5 decisions at 4096 envs) at once -- the reference's learner takes 1024 decisions per step for the same
    reason (driver.py:144-147, BATCH_SIZE).  Returns (loss sum as a float, number of decisions)."""
    agents, tasks, action, mask, adv = res[0], res[1], res[2], res[3], res[6]
    n, total = agents.shape[0], 0.0
    for lo in range(0, n, minibatch):
        hi = min(n, lo + minibatch)
        logp = net(tasks[lo:hi], agents[lo:hi], mask[lo:hi])
        loss = -(torch.gather(logp, 1, action[lo:hi]) * adv[lo:hi].detach()).sum()
        (loss * scale).backward()
        total += float(loss.detach())
    return total, n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--total-envs", type=int, default=256)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--embedding", type=int, default=128)
    ap.add_argument("--agents", type=int, nargs=2, default=(10, 20))
    ap.add_argument("--tasks", type=int, nargs=2, default=(20, 50))
    ap.add_argument("--learner", choices=("all_reduce", "rank0"), default="all_reduce")
    ap.add_argument("--lr", type=float, default=1e-5)
    ap.add_argument("--minibatch", type=int, default=8192, help="decisions per forward / backward of the learner (gradients accumulate)")
    ap.add_argument("--precision", choices=("fp32", "fp16", "bf16"), default="fp32", help="arithmetic of the ROLLOUT policy (the learner is fp32)")
    ap.add_argument("--dump", default=None, help="directory for per-rank result files (tests)")
    args = ap.parse_args()
    ctx = DistContext.from_env()
    torch.manual_seed(1234 + ctx.rank)       # deliberately different initial weights per rank: broadcast_weights must fix that
    sr = ShardedRunner(args.total_envs, ctx=ctx, net_factory=lambda: AttentionNet(6, 5, args.embedding), base_seed=7,
                       rollout_precision=args.precision)
    sr.runner.keep_greedy_record = bool(args.dump)
    sr.broadcast_weights(src=0)
    net = sr.runner.localNetwork
    opt = torch.optim.Adam(net.parameters(), lr=args.lr)
    rng = np.random.default_rng(0)
    log = []
    for rnd in range(args.rounds):
        # driver.py:114-115: one (agents_num, tasks_num) per round, the same on every rank
        shape = torch.tensor([int(rng.integers(args.agents[0], args.agents[1] + 1)), int(rng.integers(args.tasks[0], args.tasks[1] + 1))],
                             device=ctx._coll_device())
        if ctx.active:
            dist.broadcast(shape, src=0)
        A, T = int(shape[0]), int(shape[1])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res, metrics, info = sr.job(rnd, A, T)
        torch.cuda.synchronize()
        if args.dump:
            rec = sr.runner.last["rec"]                      # the sampled episodes' record [n_steps, B_local, ...]
            n = sr.runner.last["n_steps"]
            np.savez(os.path.join(args.dump, f"round{rnd}_rank{ctx.rank}.npz"), A=A, T=T, lo=sr.lo, hi=sr.hi,
                     returns=info["returns"].cpu().numpy(), summary=sr.runner.last["summary"].cpu().numpy(),
                     greedy_summary=sr.runner.last["greedy_summary"].cpu().numpy(),
                     **{k: v[:n].cpu().numpy() for k, v in rec.items()})
        t_roll = time.perf_counter() - t0
        opt.zero_grad()
        if args.learner == "all_reduce":
            loss, n_loc = reinforce_backward(net, res, minibatch=args.minibatch)
            n_all = sr.all_reduce_gradients(net, n_loc)
        else:
            gathered = sr.gather_experience(res, dst=0)
            n_all = 0
            if ctx.rank == 0:
                n_all = gathered[0].shape[0]
                loss, _ = reinforce_backward(net, gathered, scale=1.0 / n_all, minibatch=args.minibatch)
        if args.learner == "all_reduce" or ctx.rank == 0:
            torch.nn.utils.clip_grad_norm_(net.parameters(), max_norm=1000.0)       # driver.py:185
            opt.step()
        if args.learner == "rank0":
            sr.broadcast_weights(src=0)
        else:
            sr.runner.set_weights(net.state_dict())                                  # refresh a low-precision shadow, if any
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        log.append(dict(round=rnd, A=A, T=T, decisions=n_all, makespan=metrics["makespan"], success_rate=metrics["success_rate"],
                        weights_checksum=sr.weights_checksum(), rollout_seconds=round(t_roll, 4), learner_seconds=round(t_all - t_roll, 4),
                        decisions_per_s=round(n_all / t_all, 1)))
    if args.dump:
        with open(os.path.join(args.dump, f"log_rank{ctx.rank}.json"), "w") as f:
            json.dump(log, f)
    if ctx.rank == 0:
        for l in log:
            print(json.dumps(l), flush=True)
    sr.close()
    ctx.shutdown()


if __name__ == "__main__":
    main()
