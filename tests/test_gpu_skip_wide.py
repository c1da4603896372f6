"""precision='exact' beyond 128 features: block skipping on the wide screen (csrc/exact_skip_wide.hpp) -- euclidean, resident rows
from their second epoch on.  The checker is the float32 kernel: identical ids in every epoch, identical codebooks.  GPU only."""
import numpy as np
import pytest

from oracle import som_oracle as O

pytestmark = pytest.mark.gpu
F32 = np.float32


def engine(X, Y, D, **kw):
    from xpysom_dask_amd.engine import HipEngine
    return HipEngine(X, Y, D, **kw)


def train_both(X, Y, D, data, T, w, sigma0=None, check_each=True, **kw):
    f = engine(X, Y, D, precision="f32", **kw)
    x = engine(X, Y, D, precision="exact", **kw)
    for e in (f, x):
        e.set_weights(w)
        e.set_data(data)
    shares = []
    sigma0 = sigma0 or min(X, Y) / 2.0
    for t in range(T):
        sig, eta = O.exponential_decay(sigma0, 1.0, t, T), O.exponential_decay(0.5, 0.01, t, T)
        r0, t0 = x.exact_skip_stats()
        f.epoch_accumulate(sig, eta, True)
        x.epoch_accumulate(sig, eta, True)
        r1, t1 = x.exact_skip_stats()
        shares.append((r1 - r0) / max(1, t1 - t0))
        if check_each:
            a, b = f.epoch_fetch()[2], x.epoch_fetch()[2]
            assert np.array_equal(a, b), (t, int((a != b).sum()), np.flatnonzero(a != b)[:8])
        f.epoch_merge()
        x.epoch_merge()
    wf, wx = f.get_weights(), x.get_weights()
    stats = (x.exact_stats(), x.exact_resident_stats())
    f.close()
    x.close()
    return wf, wx, shares, stats


@pytest.mark.parametrize("X,Y,D,n", [(64, 64, 200, 12000), (128, 64, 784, 9000), (64, 72, 129, 5000), (96, 96, 800, 6000)])
def test_wide_block_skipping_trains_the_float32_map(monkeypatch, X, Y, D, n):
    monkeypatch.setenv("SOM_EXACT_SKIP", "2")
    from xpysom_dask_amd.synthetic import gaussian_blobs
    data = gaussian_blobs(n, D, seed=D)
    w = O.default_codebook(X, Y, D, 5).astype(F32)
    T = 8
    wf, wx, shares, (ex, res) = train_both(X, Y, D, data, T, w)
    assert np.array_equal(wf, wx)
    assert shares[0] == 1.0 and res[0] == T - 1              # (the first epoch has no last BMU; every later one runs under a plan)
    assert min(shares[2:]) < 1.0, shares                     # (maps of 64 to 144 groups: modest skipping -- the full-size shard is in test_gpu_fullsize.py)
    assert ex[1] <= ex[0] // 100
    print("executed shares:", [round(v, 3) for v in shares])


def test_wide_block_skipping_default_switches_and_cosine(monkeypatch):
    """Default: on from 4 096 units for the euclidean distance; the cosine distance (configs[4]) runs every block
    (tools/skip_probe_c5.py: nothing to skip there)."""
    monkeypatch.delenv("SOM_EXACT_SKIP", raising=False)
    from xpysom_dask_amd.synthetic import gaussian_blobs
    X, Y, D, n, T = 64, 64, 256, 16384, 6
    data = np.abs(gaussian_blobs(n, D, seed=3))
    w = np.abs(O.default_codebook(X, Y, D, 2)).astype(F32)
    wf, wx, shares, (ex, res) = train_both(X, Y, D, data, T, w)
    assert np.array_equal(wf, wx) and res[0] >= 1 and min(shares) < 1.0, (shares, res)
    wf, wx, shares, (ex, res) = train_both(X, Y, D, data, 3, w, distance="cosine")
    assert np.array_equal(wf, wx) and res[0] == 0 and min(shares) == 1.0


@pytest.mark.parametrize("env", [{"SOM_EXACT_RESORT": "1"}, {"SOM_EXACT_RESORT": "1000"}, {"SOM_EXACT_PASS_ROWS": "2048"},
                                 {"SOM_EXACT_PASS_ROWS": "1024", "SOM_EXACT_RESORT": "3"}])
def test_wide_block_skipping_orders_and_passes(monkeypatch, env):
    """A fresh order every epoch, one order kept forever, several passes per epoch: same ids."""
    monkeypatch.setenv("SOM_EXACT_SKIP", "2")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    from xpysom_dask_amd.synthetic import gaussian_blobs
    X, Y, D, n, T = 64, 64, 160, 7000, 6
    data = gaussian_blobs(n, D, seed=12)
    w = O.default_codebook(X, Y, D, 9).astype(F32)
    wf, wx, shares, _ = train_both(X, Y, D, data, T, w)
    assert np.array_equal(wf, wx)


def test_wide_block_skipping_with_ties_nan_rows_and_a_moved_codebook(monkeypatch):
    monkeypatch.setenv("SOM_EXACT_SKIP", "2")
    rng = np.random.RandomState(1)
    X, Y, D, n = 64, 64, 144, 4000
    w = rng.randint(-2, 3, size=(X * Y, D)).astype(F32)
    w[3000] = w[17]
    w[900] = w[17]
    data = rng.randint(-2, 3, size=(n, D)).astype(F32)
    data[5] = 0
    data[6] = w[17]
    data[9] = 1e4
    bad = data.copy()                                        # NaN / infinite rows poison the codebook at the first merge: their own run
    bad[7] = np.nan
    bad[8, 3] = np.inf
    f = engine(X, Y, D, precision="f32")
    x = engine(X, Y, D, precision="exact")
    for e in (f, x):
        e.set_weights(w.reshape(X, Y, D))
        e.set_data(bad)
    for t in range(3):
        f.epoch_accumulate(3.0, 0.3, True)
        x.epoch_accumulate(3.0, 0.3, True)
        a, b = f.epoch_fetch()[2], x.epoch_fetch()[2]
        assert np.array_equal(a, b), (t, np.flatnonzero(a != b)[:8])
        f.epoch_merge()
        x.epoch_merge()
    assert np.array_equal(f.get_weights(), x.get_weights(), equal_nan=True)
    for e in (f, x):
        e.set_weights(w.reshape(X, Y, D))
        e.set_data(data)
    for t in range(5):
        if t == 3:                                           # a codebook replaced between epochs: the last BMUs say nothing (a valid, useless bound)
            w2 = rng.randn(X * Y, D).astype(F32)
            f.set_weights(w2)
            x.set_weights(w2)
        f.epoch_accumulate(3.0, 0.3, True)
        x.epoch_accumulate(3.0, 0.3, True)
        a, b = f.epoch_fetch()[2], x.epoch_fetch()[2]
        assert np.array_equal(a, b), (t, np.flatnonzero(a != b)[:8])
        assert np.array_equal(f.bmu(data[:500]), x.bmu(data[:500]))       # queries in between (no plan for them beyond 128 features)
        f.epoch_merge()
        x.epoch_merge()
    assert np.array_equal(f.get_weights(), x.get_weights())
    f.close()
    x.close()
