"""Roofline arithmetic of the hot path.

Algorithmic bytes per env step: W = 2*S + O + 4 -- what the lockstep kernel k_step really moves per
decision.  The persistent kernel keeps the record in LDS and is bound by VALU issue instead: its utilisation is computed
from SQ counters (profiles/counters.json, written by tools/collect_profile.py from the rocprofv3 CSVs under profiles/).
"""
import json
import os


def state_bytes(A, T):
    """Canonical compact state of one env: 64 B globals + 48 B/agent + 96 B/task."""
    return 64 + 48 * A + 96 * T


def observation_bytes(A, T):
    """f32[A,6] + f32[T+1,5] + u8[T+1] (worker.py:57-68)."""
    return 24 * A + 21 * (T + 1)


def algorithmic_bytes_per_step(A, T):
    """read state + write state + write observation + read action."""
    return 2 * state_bytes(A, T) + observation_bytes(A, T) + 4


HBM_PEAK_BYTES_PER_S = 8.0e12  # MI355X HBM3E peak (MI355X_MICROARCH.md)
N_SIMD = 256 * 4               # 256 CUs x 4 SIMDs; one wave64 VALU instruction occupies a SIMD for 4 clocks
PEAK_CLOCK_HZ = 2.4e9          # peak engine clock (MI355X_MICROARCH.md per-instruction table: "256 CU x 2.4 GHz")

_COUNTERS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "counters.json")


def load_counters(key):
    """Per-decision / per-launch PMC averages of one profiled workload, or None when none is committed for it."""
    try:
        with open(_COUNTERS) as f:
            return json.load(f).get(key)
    except (OSError, ValueError):
        return None
