"""Synthetic workload of the benchmark: seeded Gaussian blobs, float32."""
import numpy as np


def gaussian_blobs(n_rows, n_features, seed=1234, centres=64, spread=3.0):
    rng = np.random.default_rng(seed)
    c = rng.normal(0.0, spread, size=(centres, n_features))
    lab = rng.integers(0, centres, size=n_rows)
    x = c[lab] + rng.normal(0.0, 1.0, size=(n_rows, n_features))
    return x.astype(np.float32)
