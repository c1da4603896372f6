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
    assert shares[0] > 0.95 and res[0] == T                  # (mode 2: the first epoch under the scout's plan -- a random codebook keeps (nearly) everything)
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


def test_wide_scout_queries_streams_and_first_epochs(monkeypatch):
    """Beyond 128 features rows WITHOUT a last BMU -- query rows, streamed chunks, a row set's first epoch -- get a pseudo last
    BMU from the scout (the plain wide kernel on the centroid image, then over the nearest group's units): same ids as float32,
    the launches ran under a plan, and on a trained map most blocks were skipped."""
    monkeypatch.setenv("SOM_EXACT_SKIP", "2")
    from xpysom_dask_amd.synthetic import gaussian_blobs
    X, Y, D, n, T = 96, 64, 300, 14000, 6
    data = gaussian_blobs(n, D, seed=31)
    probe = gaussian_blobs(6000, D, seed=32, centre_seed=31)
    w = O.default_codebook(X, Y, D, 4).astype(F32)
    f = engine(X, Y, D, precision="f32")
    x = engine(X, Y, D, precision="exact")
    for e in (f, x):
        e.set_weights(w)
        e.set_data(data)
    for t in range(T):
        sig, eta = O.exponential_decay(32.0, 1.0, t, T), O.exponential_decay(0.5, 0.01, t, T)
        f.epoch_accumulate(sig, eta, True)
        x.epoch_accumulate(sig, eta, True)
        a, b = f.epoch_fetch()[2], x.epoch_fetch()[2]
        assert np.array_equal(a, b), (t, int((a != b).sum()))
        f.epoch_merge()
        x.epoch_merge()
    planned, _ = x.exact_resident_stats()
    assert planned == T                                      # (the first epoch too: the scout)
    r0, t0 = x.exact_skip_stats()
    assert np.array_equal(f.bmu(probe), x.bmu(probe))
    r1, t1 = x.exact_skip_stats()
    assert r1 - r0 < t1 - t0
    scouted, transient = x.exact_scout_stats()
    assert scouted >= 2 and transient >= 1
    outs = {}
    for name, e in (("f32", f), ("exact", x)):
        e.stream_epoch_accumulate([probe[:2500], probe[2500:2501], probe[2501:]], 2.0, 0.3, True)
        outs[name] = e.epoch_fetch(want_bmu=False)[:2]
    assert np.array_equal(outs["exact"][0], outs["f32"][0]) and np.array_equal(outs["exact"][1], outs["f32"][1])
    assert abs(f.quantization_error(probe) - x.quantization_error(probe)) <= 1e-6 * f.quantization_error(probe)
    f.close()
    x.close()
