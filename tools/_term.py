import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dcmrta_amd.batched_env import BatchedTaskEnv
from dcmrta_amd.choice import env_seeds
from dcmrta_amd.instances import generate_batch
A, T, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
inst = generate_batch(B, A, T, base_seed=5)
seeds = env_seeds(3, 0, B)
env = BatchedTaskEnv(B, A, T, device="cuda:0").load_instances(**inst)
def run(budget):
    env.reset(seeds, observe=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    steps = env.rollout_random(episodes=1, max_decisions=budget)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1), steps.cpu().numpy()
ms, n = run(-1)
for rep in range(2):
    ms_full, n = run(-1)
    ms_part, n2 = run(np.maximum(n - 1, 0).astype(np.int64))
    ms_half, n3 = run((n // 2).astype(np.int64))
    print(f"{A}A/{T}T B={B}: full {ms_full:.3f} ms ({n.sum()} decisions, {n.mean():.0f}/env), all-but-last {ms_part:.3f} ms, half {ms_half:.3f} ms "
          f"-> per-decision {ms_half / n3.sum() * 1e6 * B / 1:.1f} ns*B; terminal+last = {(ms_full - ms_part) / ms_full * 100:.1f} % of the launch")
