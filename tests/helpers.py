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
