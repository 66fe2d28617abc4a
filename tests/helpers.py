import numpy as np


def rel_rms(a, b):
    """RMS(a - b) / RMS(b): the north-star tolerance is 1e-5 relative RMS."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.sqrt(np.mean(b * b))
    num = np.sqrt(np.mean((a - b) ** 2))
    if den == 0.0:
        return 0.0 if num == 0.0 else np.inf
    return num / den


TOL = 1e-5


def mix_matches(got, want, tol=TOL, abs_floor=1e-8):
    """1e-5 relative RMS, or -- for ring-out tails far below audibility (|x| ~ 1e-12 .. denormal) where a
    relative measure is meaningless -- an absolute RMS error under 1e-8."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    err = np.sqrt(np.mean((got - want) ** 2))
    return err <= abs_floor or err <= tol * np.sqrt(np.mean(want * want))
