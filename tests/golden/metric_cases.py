"""Deterministic inputs shared by make_golden_metrics.py (run against the reference) and tests/test_metrics.py."""
import numpy as np


def metric_case(seed: int):
    rng = np.random.default_rng(1100 + seed)
    n = int(rng.integers(30, 600))
    y = np.zeros(n, dtype=int)
    v = 0
    for i in range(n):
        if rng.random() < 0.05:
            v = 1 - v
        y[i] = v
    if y.sum() in (0, n):
        y[n // 2] = 1 - y[n // 2]
    probs = np.clip(0.5 + (y - 0.5) * rng.uniform(0.1, 0.6) + rng.normal(0, 0.25, n), 0, 1)
    P = np.clip(probs[:, None] + rng.normal(0, 0.1, (n, 7)), 0, 1).astype(np.float32)
    return y, P
