"""Algorithmic bytes per env step: W = 2*S + O + 4."""


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
