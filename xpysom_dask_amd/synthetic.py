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


VARIANTS = ("blobs", "overlap", "manifold", "heavy", "normal")


def variant(kind, n_rows, n_features, seed=1234):
    """Row sets between "64 well separated blobs" and "no structure at all" (bench.py `data_variants`, the policy tests):
      blobs     the benchmark's rows (gaussian_blobs)
      overlap   1 024 centres ~ N(0, 9 I) with noise of the SAME spread (sigma 3): clusters that overlap their neighbours
      manifold  rows on a 2-D sheet embedded in n_features dimensions (random Fourier features of (u, v) in the unit
                square, amplitude 3) + N(0, 0.05^2 I): the structure a SOM is meant to unfold, no clusters at all
      heavy     64 blobs whose sizes follow a power law (size ~ rank^-1.2: the largest holds a fifth of the rows)
      normal    N(0, I): no structure"""
    rng = np.random.default_rng(seed)
    if kind == "blobs":
        return gaussian_blobs(n_rows, n_features, seed=seed, centre_seed=1234)
    if kind == "normal":
        return rng.standard_normal((n_rows, n_features)).astype(np.float32)
    if kind == "overlap":
        c = np.random.default_rng(1234).normal(0.0, 3.0, size=(1024, n_features))
        lab = rng.integers(0, 1024, size=n_rows)
        return (c[lab] + rng.normal(0.0, 3.0, size=(n_rows, n_features))).astype(np.float32)
    if kind == "manifold":
        g = np.random.default_rng(1234)
        a, b = g.normal(0.0, 2.5, size=n_features), g.normal(0.0, 2.5, size=n_features)
        ph = g.uniform(0.0, 2.0 * np.pi, size=n_features)
        out = np.empty((n_rows, n_features), dtype=np.float32)
        for lo in range(0, n_rows, 1 << 18):
            hi = min(n_rows, lo + (1 << 18))
            u, v = rng.uniform(size=(hi - lo, 1)), rng.uniform(size=(hi - lo, 1))
            out[lo:hi] = 3.0 * np.sin(u * a[None, :] + v * b[None, :] + ph[None, :]) + rng.normal(0.0, 0.05, size=(hi - lo, n_features))
        return out
    if kind == "heavy":
        c = np.random.default_rng(1234).normal(0.0, 3.0, size=(64, n_features))
        p = np.arange(1, 65, dtype=np.float64) ** -1.2
        lab = rng.choice(64, size=n_rows, p=p / p.sum())
        return (c[lab] + rng.normal(0.0, 1.0, size=(n_rows, n_features))).astype(np.float32)
    raise ValueError("unknown data variant %r (one of %s)" % (kind, ", ".join(VARIANTS)))
