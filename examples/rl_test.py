#!/usr/bin/env python3
"""RL_test.py of the reference (RL_test.py:1-51), batched: evaluate a policy greedily on a test set of instances with
Worker.run_test (METHOD "LF", leader-follower, worker.py:114-157) or run_test_IS (METHOD "IA", individual selection,
worker.py:159-198) and write the per-instance result CSV the reference's plotting scripts read (RL_test.py:45-51).

The reference loops over 50 pickled envs one at a time; here all instances are one batch on the GPU.

    python examples/rl_test.py [--instances tests/golden/instances_20A50T.npz] [--checkpoint checkpoint.pth] [--method LF|IA]
                               [--out REINFORCE_LF.csv]

--instances: npz with depot[N,2], task_xy[N,T,2], req[N,T], dur[N,T], A (dcmrta_amd.instances.load_instances_npz; the
shipped fixture holds the reference's testSet_20A_50T_CONDET instances) -- unpickled reference envs can be converted with
dcmrta_amd.instances.batch_from_dicts.  --checkpoint: a checkpoint of the REFERENCE's AttentionNet
({'model': state_dict}, RL_test.py:29-30), mapped onto the stand-in by load_reference_state_dict; without it the net is
randomly initialised (the reference's own checkpoint blob is not distributed).
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dcmrta_amd.instances import load_instances_npz  # noqa: E402
from dcmrta_amd.policy import AttentionNet, load_reference_state_dict  # noqa: E402
from dcmrta_amd.runner import METRIC_KEYS, BatchedRunner  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--instances", default=os.path.join(ROOT, "tests", "golden", "instances_20A50T.npz"))
    ap.add_argument("--checkpoint", default=None)
    ap.add_argument("--method", choices=["LF", "IA"], default="LF")
    ap.add_argument("--out", default=None)
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    torch.manual_seed(a.seed)
    inst, A = load_instances_npz(a.instances)
    N = inst["req"].shape[0]
    runner = BatchedRunner(n_envs=N, device=a.device, net_factory=lambda: AttentionNet(6, 5, 128))   # AGENT_INPUT_DIM, TASK_INPUT_DIM, EMBEDDING_DIM
    if a.checkpoint:
        ck = torch.load(a.checkpoint, map_location="cpu")                                           # RL_test.py:29-30
        load_reference_state_dict(runner.localNetwork, ck["model"])
    m = runner.run_test(inst, n_agents=A, individual_selection=(a.method == "IA"))                    # RL_test.py:44-47
    out = a.out or f"REINFORCE_{a.method}.csv"
    with open(out, "w") as f:                                                                        # RL_test.py:48-51 (pandas layout)
        f.write("," + ",".join(METRIC_KEYS) + "\n")
        for i in range(N):
            f.write(str(i) + "," + ",".join(repr(float(m[k][i])) for k in METRIC_KEYS) + "\n")
    print(f"{N} instances, METHOD {a.method}: makespan {np.mean(m['makespan']):.3f} +- {np.std(m['makespan']):.3f}, "
          f"success_rate {np.mean(m['success_rate']):.3f} -> {out}")


if __name__ == "__main__":
    main()
