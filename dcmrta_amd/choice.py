"""Keyed choice protocol shared by the HIP kernels, the oracle and the golden generator.

The reference draws the leader from the global numpy RNG (worker.py:54) and the followers
from Generator.choice (env/task_env.py:50-55,331); neither stream can be reproduced on a
GPU, so parity is defined on a counter-based protocol instead:

    mix64          splitmix64 finaliser
    env_seed(b,e)  = mix64(b + GAMMA*(e+1))              per-env seed from a base seed
    key_1(s,d)     = mix64(s + GAMMA*(d+1))              d = running decision counter of the env
    key_{k+1}      = mix64(key_k + GAMMA)
    draw(s,d,slot) = 32-bit word `slot` of the stream hi(key_1), lo(key_1), hi(key_2), lo(key_2), ...
                     slot 0 leader, 1 action, 2+j follower j
    below(r, n)    = (r * n) >> 32                       multiply-high range reduction (no division)

    leader    = group[below(draw(s,d,0), len(group))]    group in ascending agent-id order
    action    = valid[below(draw(s,d,1), len(valid))]    uniform-random policy only
    followers = k successive rest.pop(below(draw(s,d,2+j), len(rest)))

Every draw is a pure function of (seed, d, slot): observe() and step() recompute the same
leader without carrying RNG state, and a policy that ignores slot 1 does not shift the others.
One 64-bit mix serves leader + action of a decision; a second one is only needed when
followers are drawn (one wave-uniform computation on the scalar unit of the GPU).
"""
import numpy as np

GAMMA = 0x9E3779B97F4A7C15
_M64 = (1 << 64) - 1


def mix64(z: int) -> int:
    z &= _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def env_seed(base: int, e: int) -> int:
    return mix64(base + GAMMA * (e + 1))


def draw(seed_e: int, d: int, slot: int) -> int:
    """32-bit word `slot` of the decision's stream."""
    key = mix64(seed_e + GAMMA * (d + 1))
    for _ in range(slot // 2):
        key = mix64(key + GAMMA)
    return (key >> 32) if slot % 2 == 0 else (key & 0xFFFFFFFF)


def below(r: int, n: int) -> int:
    return (r * n) >> 32


def env_seeds(base: int, first: int, count: int) -> np.ndarray:
    """Vectorised env_seed(base, first .. first+count-1) as uint64[count]."""
    with np.errstate(over="ignore"):
        e = np.arange(first + 1, first + count + 1, dtype=np.uint64)
        z = np.uint64(base & _M64) + np.uint64(GAMMA) * e
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))
