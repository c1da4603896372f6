"""Synthetic workload of the benchmark: seeded Gaussian blobs, float32."""
import numpy as np


def gaussian_blobs(n_rows, n_features, seed=1234, centres=64, spread=3.0, centre_seed=None):
    """`centre_seed` (default: `seed`) fixes the blob centres independently of the rows drawn, so that shards
    and chunks generated from different seeds are samples of ONE mixture."""
    rng = np.random.default_rng(seed)
    c = rng.normal(0.0, spread, size=(centres, n_features))
    if centre_seed is not None and centre_seed != seed:
        c = np.random.default_rng(centre_seed).normal(0.0, spread, size=(centres, n_features))
    lab = rng.integers(0, centres, size=n_rows)
    x = c[lab] + rng.normal(0.0, 1.0, size=(n_rows, n_features))
    return x.astype(np.float32)
