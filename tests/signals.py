"""Seeded synthetic audio used by tests and bench (BASELINE.md §4 generator)."""

import numpy as np


def make_streams(n: int, T: int, seed: int = 1234, first_stream: int = 0) -> np.ndarray:
    """[n, T, 512] float32.  Stream s is seeded default_rng(seed + s); even s: N(0, 0.02^2) noise,
    odd s: 0.3*(0.4 sin150 + 0.3 sin300 + 0.2 sin600 Hz) + N(0, 0.03^2); clipped to [-1, 1]."""
    out = np.empty((n, T * 512), np.float32)
    t = np.arange(T * 512, dtype=np.float64) / 16000.0
    for i in range(n):
        s = first_stream + i
        rng = np.random.default_rng(seed + s)
        if s % 2 == 0:
            x = 0.02 * rng.standard_normal(t.size)
        else:
            x = 0.3 * (0.4 * np.sin(2 * np.pi * 150 * t) + 0.3 * np.sin(2 * np.pi * 300 * t)
                       + 0.2 * np.sin(2 * np.pi * 600 * t)) + 0.03 * rng.standard_normal(t.size)
        out[i] = np.clip(x, -1.0, 1.0).astype(np.float32)
    return out.reshape(n, T, 512)
