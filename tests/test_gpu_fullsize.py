"""BASELINE.json's full-size configurations through the C ABI, checked by size-independent
properties (the oracle cannot run a 1 Mi x 65,536 x 128 epoch in seconds):

  * BMU optimality on a random subset of rows (float64 distances on the host);
  * the update is the exact separable transform of the per-unit segment sums:
    den == (Px (x) Py) c and num == (Px (x) Py) S recomputed in float64 from the engine's own BMUs;
  * shard linearity: accumulate(first half) + accumulate(second half) == accumulate(all);
  * merge: W' == num/den where den != 0 and W' == W elsewhere.
"""
import numpy as np
import pytest

from oracle import som_oracle as O
from tests.test_gpu_parity import bf16_misses_are_near_best, engine, near_tie_mask, rel_err

pytestmark = pytest.mark.gpu
F32 = np.float32


def gaussian_tables(X, Y, sigma, eta, std_coeff=0.5):
    d = 2 * std_coeff ** 2 * sigma ** 2
    ix, iy = np.arange(X, dtype=np.float64), np.arange(Y, dtype=np.float64)
    Px = np.exp(-(ix[:, None] - ix[None, :]) ** 2 / d) * eta
    Py = np.exp(-(iy[:, None] - iy[None, :]) ** 2 / d)
    return Px, Py


def separable_update(data, bmu, X, Y, sigma, eta):
    """float64 restatement of num/den from the segment sums (exact algebra of xpysom.py:434-441)."""
    K, D = X * Y, data.shape[1]
    S = np.zeros((K, D))
    np.add.at(S, bmu, data.astype(np.float64))
    c = np.bincount(bmu, minlength=K).astype(np.float64)
    Px, Py = gaussian_tables(X, Y, sigma, eta)
    num = np.einsum("ia,jb,abd->ijd", Px, Py, S.reshape(X, Y, D), optimize=True).reshape(K, D)
    den = (Px @ c.reshape(X, Y) @ Py.T).reshape(K)
    return num, den


@pytest.mark.parametrize("cfg", [
    dict(name="C2", X=64, Y=64, D=32, N=100_000, precision="f32"),       # BASELINE configs[1]
    dict(name="C3", X=256, Y=256, D=128, N=1 << 20, precision="bf16"),    # BASELINE configs[2], the throughput mode
    dict(name="C3-exact", X=256, Y=256, D=128, N=1 << 20, precision="exact"),   # ... and the mode bench.py times
], ids=lambda c: c["name"])
def test_full_size_epoch_properties(cfg):
    X, Y, D, N, precision = cfg["X"], cfg["Y"], cfg["D"], cfg["N"], cfg["precision"]
    K = X * Y
    data = O.gaussian_blobs(N, D, seed=1234)
    w = O.default_codebook(X, Y, D, 1234).astype(F32)
    sigma, eta = min(X, Y) / 4.0, 0.3

    e = engine(X, Y, D, precision=precision)
    e.set_weights(w)
    e.set_data(data)
    e.epoch_accumulate(sigma, eta, True)
    num, den, bmu = e.epoch_fetch()
    assert bmu.min() >= 0 and bmu.max() < K

    # 1. BMU optimality on a subset
    rs = np.random.RandomState(0)
    pick = rs.choice(N, 1536, replace=False)
    wf = w.reshape(K, D)
    ref = O.bmu_ids(data[pick], wf)
    bad = np.flatnonzero(bmu[pick] != ref)
    if precision in ("f32", "exact"):
        assert near_tie_mask(data[pick][bad], wf, tol=1e-5).all()
    else:
        assert bf16_misses_are_near_best(data[pick], wf, bmu[pick], bad)

    # 2. the accumulators are the separable transform of the segment sums of the engine's own BMUs
    onum, oden = separable_update(data, bmu, X, Y, sigma, eta)
    assert rel_err(den, oden) < 1e-5
    assert rel_err(num, onum) < 1e-5

    # 3. merge
    e.epoch_merge()
    w1 = e.get_weights()
    live = den != 0
    np.testing.assert_array_equal(w1[live], (num[live] / den[live, None]).astype(F32))
    np.testing.assert_array_equal(w1[~live], wf[~live])

    # 4. shard linearity (what the all-reduce relies on), from the same codebook
    tot_num, tot_den = np.zeros_like(num, dtype=np.float64), np.zeros_like(den, dtype=np.float64)
    half = N // 2
    for lo, hi in ((0, half), (half, N)):
        e.set_weights(w)
        e.set_data(data[lo:hi])
        e.epoch_accumulate(sigma, eta, True)
        pn, pd, pb = e.epoch_fetch()
        if precision in ("f32", "exact"):
            assert np.array_equal(pb, bmu[lo:hi])       # a row's BMU does not depend on its shard
        else:
            # bf16: the positivity offset B = max|x~| max|w~| is a property of the shard, so a
            # near-tie may round the other way; any such row must still hold a near-best unit
            moved = np.flatnonzero(pb != bmu[lo:hi])
            assert len(moved) < 0.01 * (hi - lo)
            assert bf16_misses_are_near_best(data[lo:hi][moved[:512]], wf, pb[moved[:512]], np.arange(min(512, len(moved))))
        tot_num += pn
        tot_den += pd
    tol = 2e-6 if precision in ("f32", "exact") else 2e-3   # bf16: the few moved rows above
    assert rel_err(tot_num, num) < tol and rel_err(tot_den, den) < tol


def test_full_size_c3_exact_skip_equals_float32():
    """configs[2] at FULL size in the mode bench.py times: 1 Mi resident rows, the benchmark's 12-epoch schedule from the
    seeded codebook, precision='exact' with block skipping in its default mode against precision='f32' -- the BMUs of
    EVERY row in EVERY epoch bit for bit, hence the same trained codebook; and from the fourth epoch on most blocks of
    the distance GEMM were proved empty, not run (csrc/exact_skip.hpp).  ~1.5 s of float32 kernel per epoch pair."""
    from xpysom_dask_amd.synthetic import gaussian_blobs
    X = Y = 256; D = 128; N = 1 << 20; T = 12
    data = gaussian_blobs(N, D, seed=1234, centre_seed=1234)         # bench.py's rows (workload_rows("c3", ...))
    w = O.default_codebook(X, Y, D, 1234).astype(F32)
    f = engine(X, Y, D, precision="f32"); x = engine(X, Y, D, precision="exact")
    for e in (f, x):
        e.set_weights(w); e.set_data(data)
    shares = []
    for t in range(T):
        sig, eta = O.exponential_decay(128.0, 1.0, t, T), O.exponential_decay(0.5, 0.01, t, T)
        r0, t0 = x.exact_skip_stats()
        f.epoch_accumulate(sig, eta, True); x.epoch_accumulate(sig, eta, True)
        r1, t1 = x.exact_skip_stats()
        shares.append((r1 - r0) / (t1 - t0))
        bf, bx = f.epoch_fetch()[2], x.epoch_fetch()[2]
        assert np.array_equal(bf, bx), "epoch %d: %d of %d rows differ" % (t, int((bf != bx).sum()), N)
        f.epoch_merge(); x.epoch_merge()
    assert np.array_equal(f.get_weights(), x.get_weights())
    rows, fb, _ = x.exact_stats()
    assert rows == N * T and fb == 0
    assert shares[0] == 1.0 and max(shares[4:]) < 0.5 and min(shares[4:]) < 0.15, shares
    f.close(); x.close()


def test_default_constructed_xpysom_trains_through_the_screen_and_equals_float32():
    """The drop-in's DEFAULT precision is 'exact' (the reference's kwargs and nothing else): a default-constructed
    XPySom(256, 256, 128) trains through the MFMA screen (rows screened > 0, none through the float32 fallback kernel) and
    ends on the codebook precision='f32' ends on, bit for bit (xpysom.py:73-82)."""
    from xpysom_dask_amd import XPySom
    from xpysom_dask_amd.synthetic import gaussian_blobs
    X = Y = 256; D = 128; N = 65536; T = 4
    data = gaussian_blobs(N, D, seed=7)
    a = XPySom(X, Y, D, random_seed=1234)
    assert a._precision == "exact"
    a.train(data, T)
    rows, fb, _ = a._engine().exact_stats()
    assert rows == N * T and fb == 0
    b = XPySom(X, Y, D, random_seed=1234, precision="f32").train(data, T)
    assert np.array_equal(a._weights, b._weights)
    assert a.winner(data[:2000]) == b.winner(data[:2000])


def test_full_size_quantization_error_and_winner_c2():
    """C2 map: quantization_error == mean distance to the winner's codebook row, winner == argmin."""
    from xpysom_dask_amd import XPySom
    X, Y, D, N = 64, 64, 32, 100_000
    data = O.gaussian_blobs(N, D, seed=99)
    som = XPySom(X, Y, D, random_seed=1234, decay_function="linear")
    som.train(data, 3)
    w = som._weights.reshape(-1, D)
    ids = som._winner_ids(data, quantization=True)
    qe = som.quantization_error(data)
    direct = np.linalg.norm(data.astype(np.float64) - w[ids], axis=1).mean()
    assert abs(qe - direct) < 1e-5 * direct
    rs = np.random.RandomState(1)
    pick = rs.choice(N, 2048, replace=False)
    assert abs(O.quantization_error(data[pick], som._weights) -
               np.linalg.norm(data[pick].astype(np.float64) - w[ids[pick]], axis=1).mean()) < 1e-4
    # top-2 by VALUE: the reference takes them from an unstable argsort, so among exactly equal
    # float32 distances (frequent on a smooth map after the sqrt) the ids are not defined, the values are
    b1, b2 = som._upload_weights().bmu_top2(data[pick])
    d = O.dist_euclid(data[pick], w)
    two = np.sort(d, axis=1)[:, :2]
    rows = np.arange(len(pick))
    # (to 5e-6: the oracle's sgemm on THIS host need not sum in the order the golden vectors' host did,
    #  and sqrt(a - b) amplifies the cancellation in the squared distance)
    np.testing.assert_allclose(d[rows, b1], two[:, 0], rtol=5e-6)
    np.testing.assert_allclose(d[rows, b2], two[:, 1], rtol=5e-6)
    assert (b1 != b2).all()
    assert abs(som.topographic_error(data[pick]) - O.topographic_error(data[pick], som._weights)) < 0.05


def test_full_size_c5_shard_properties():
    """BASELINE configs[4] (512 x 512 map, 784 features, cosine + mexican_hat, bf16): one GPU's WHOLE shard
    (250 000 of the 2 M rows) through the wide kernel, checked by size-independent properties -- every BMU is the
    best or a near-best cosine match, the denominator and a random set of numerator columns are the
    two-term separable mexican-hat transform of the segment sums of the engine's own BMUs, the
    merge is num/den, and two half-shards add up to the whole."""
    X = Y = 512
    D, N = 784, 250000
    K = X * Y
    rs = np.random.RandomState(5)
    data = np.abs(O.gaussian_blobs(N, D, seed=4321))
    data /= np.linalg.norm(data, axis=1, keepdims=True)
    w = np.abs(rs.rand(K, D).astype(F32))
    sigma, eta = 40.0, 0.3
    e = engine(X, Y, D, precision="bf16", distance="cosine", neighborhood="mexican_hat")
    e.set_weights(w.reshape(X, Y, D))
    e.set_data(data)
    e.epoch_accumulate(sigma, eta, True)
    num, den, bmu = e.epoch_fetch()
    assert bmu.min() >= 0 and bmu.max() < K

    # 1. cosine optimality on a subset (float32 similarities are ample against the 2^-7 bf16 bound)
    pick = rs.choice(N, 1024, replace=False)
    wn = w / np.linalg.norm(w, axis=1, keepdims=True)
    sim = data[pick] @ wn.T
    assert (sim[np.arange(len(pick)), bmu[pick]] >= sim.max(1) - 2.0 ** -7).all()
    del sim, wn

    # 2. separable two-term mexican hat: h = ex*ey*(1 - 2dx^2/d - 2dy^2/d) = A (x) ey + ex (x) B
    d = 2 * 0.5 ** 2 * sigma ** 2
    ix = np.arange(X, dtype=np.float64)
    dx2 = (ix[:, None] - ix[None, :]) ** 2
    ex = np.exp(-dx2 / d)
    A, B = ex * (1 - 2 * dx2 / d), ex * (-2 * dx2 / d)
    c = np.bincount(bmu, minlength=K).astype(np.float64).reshape(X, Y)
    oden = eta * (A @ c @ ex.T + ex @ c @ B.T)
    assert rel_err(den.reshape(X, Y), oden) < 1e-5
    cols = rs.choice(D, 24, replace=False)
    S = np.zeros((K, len(cols)))
    np.add.at(S, bmu, data[:, cols].astype(np.float64))
    S = S.reshape(X, Y, len(cols))
    onum = eta * (np.einsum("ia,jb,abd->ijd", A, ex, S, optimize=True) + np.einsum("ia,jb,abd->ijd", ex, B, S, optimize=True))
    assert rel_err(num[:, cols], onum.reshape(K, -1)) < 1e-5

    # 3. merge
    e.epoch_merge()
    w1 = e.get_weights()
    live = den != 0
    np.testing.assert_array_equal(w1[live][:, cols], (num[live][:, cols] / den[live, None]).astype(F32))
    np.testing.assert_array_equal(w1[~live], w[~live])
    del w1

    # 4. shard linearity on the denominator and the sampled numerator columns
    tot_den = np.zeros(K)
    tot_num = np.zeros((K, len(cols)))
    moved = 0
    for lo, hi in ((0, N // 2), (N // 2, N)):
        e.set_weights(w.reshape(X, Y, D))
        e.set_data(data[lo:hi])
        e.epoch_accumulate(sigma, eta, True)
        pn, pd, pb = e.epoch_fetch()
        moved += int((pb != bmu[lo:hi]).sum())
        tot_den += pd
        tot_num += pn[:, cols]
    # the positivity offset B = max|x~| max|w~| belongs to the shard, so a near-tie may round the other way
    assert moved < 1e-3 * N
    assert rel_err(tot_den, den) < 2e-3 and rel_err(tot_num, num[:, cols]) < 2e-3


def test_full_size_c5_shard_exact_mode_equals_float32(monkeypatch):
    """configs[4]'s whole one-GPU shard (250 000 rows, 512 x 512 x 784, cosine) through precision='exact' -- the wide
    IEEE-half screen (one pass on the seeded codebook, four passes of 65 536 rows on the smooth one) + the float32 re-score on
    the tile image -- and through the float32 kernel itself: the same 250 000 BMUs, on the seeded codebook and on a smooth
    one (the early-schedule state: many candidates)."""
    X = Y = 512
    D, N = 784, 250000
    data = np.abs(O.gaussian_blobs(N, D, seed=4321))
    data /= np.linalg.norm(data, axis=1, keepdims=True)
    data = data.astype(F32)
    rs = np.random.RandomState(5)
    for state in ("seeded", "smooth"):
        if state == "seeded":
            w = np.abs(rs.rand(X, Y, D).astype(F32))
        else:
            w = np.abs(O.smooth_sheet_codebook(X, Y, D, seed=3, amplitude=0.3, centre=data[:4096].astype(np.float64).mean(0))).astype(F32)
        ids = {}
        if state == "smooth":
            monkeypatch.setenv("SOM_EXACT_PASS_ROWS", "65536")
        for p in ("exact", "f32"):
            e = engine(X, Y, D, precision=p, distance="cosine", neighborhood="mexican_hat")
            e.set_weights(w)
            e.set_data(data)
            e.epoch_accumulate(40.0, 0.3, True)
            ids[p] = e.epoch_fetch()[2]
            if p == "exact":
                rows, fb, passes = e.exact_stats()
                assert rows == N and passes == (1 if state == "seeded" else 4)
                if state == "seeded":
                    assert fb <= N // 100
            e.close()
        assert np.array_equal(ids["exact"], ids["f32"]), (state, int((ids["exact"] != ids["f32"]).sum()))


def test_full_size_wide_euclidean_shard_skips_blocks_and_equals_float32():
    """512 x 512 x 784 with the euclidean distance and the gaussian neighbourhood (the G17 family at configs[4]'s size), one
    GPU's shard of 65 536 rows, the first five epochs of a 12-epoch schedule: the wide screen runs under a plan from the second
    epoch on (csrc/exact_skip_wide.hpp), every epoch's BMUs are float32's, and from the third epoch on most blocks are skipped
    (tools/skip_probe_wide.py counted 1-14 % of the groups; profiles/r05_skip_probe_wide.txt)."""
    from xpysom_dask_amd.synthetic import gaussian_blobs
    X = Y = 512
    D, n, T = 784, 65536, 12
    data = gaussian_blobs(n, D, seed=1234, centre_seed=1234)
    rs = np.random.RandomState(1234)
    w = rs.rand(X, Y, D) * 2 - 1
    w = (w / np.linalg.norm(w, axis=-1, keepdims=True)).astype(F32)
    f = engine(X, Y, D, precision="f32")
    x = engine(X, Y, D, precision="exact")
    for e in (f, x):
        e.set_weights(w)
        e.set_data(data)
    shares = []
    for t in range(5):
        sig, eta = O.exponential_decay(256.0, 1.0, t, T), O.exponential_decay(0.5, 0.01, t, T)
        r0, t0 = x.exact_skip_stats()
        f.epoch_accumulate(sig, eta, True)
        x.epoch_accumulate(sig, eta, True)
        r1, t1 = x.exact_skip_stats()
        shares.append((r1 - r0) / (t1 - t0))
        a, b = f.epoch_fetch()[2], x.epoch_fetch()[2]
        assert np.array_equal(a, b), (t, int((a != b).sum()))
        f.epoch_merge()
        x.epoch_merge()
    assert np.array_equal(f.get_weights(), x.get_weights())
    assert shares[0] == 1.0 and max(shares[3:]) < 0.3, shares
    rows, fb, _ = x.exact_stats()
    assert fb <= rows // 100
    f.close()
    x.close()


@pytest.mark.parametrize("kind", ["overlap", "manifold", "heavy", "normal"])
def test_full_size_default_policy_is_never_much_slower_than_a_full_scan(monkeypatch, kind):
    """The plan's decisions come from measured costs (csrc/exact_policy.hpp), not from the benchmark's rows: on four other
    generators (xpysom_dask_amd/synthetic.py: 1 024 overlapping centres, a 2-D manifold, power-law cluster sizes, N(0, I)) a
    12-epoch schedule at 256 x 256 x 128 / 512 Ki rows under the default switches costs no more than the same schedule with
    SOM_EXACT_SKIP=0 (+ 5 %), no single epoch more than 12 % (a declined forecast costs 4-6 %; the rest is the box's noise), and
    trains the same codebook bit for bit."""
    import time
    from xpysom_dask_amd import synthetic
    X = Y = 256
    D, n, T = 128, 1 << 19, 12
    data = synthetic.variant(kind, n, D, seed=77)
    rs = np.random.RandomState(1234)
    w = rs.rand(X, Y, D) * 2 - 1
    w = (w / np.linalg.norm(w, axis=-1, keepdims=True)).astype(F32)
    out = {}
    for tag, skip in (("full", "0"), ("default", None)):
        if skip is None:
            monkeypatch.delenv("SOM_EXACT_SKIP", raising=False)
        else:
            monkeypatch.setenv("SOM_EXACT_SKIP", skip)
        e = engine(X, Y, D, precision="exact")
        e.set_weights(w)
        e.set_data(data)
        e.sync()
        ms = []
        for t in range(T):
            sig, eta = O.exponential_decay(128.0, 1.0, t, T), O.exponential_decay(0.5, 0.01, t, T)
            t0 = time.perf_counter()
            e.epoch(sig, eta, True)
            e.sync()
            ms.append(1e3 * (time.perf_counter() - t0))
        out[tag] = (ms, e.get_weights())
        e.close()
    assert np.array_equal(out["full"][1], out["default"][1])
    full, dflt = out["full"][0], out["default"][0]
    worst = max(a / b for a, b in zip(dflt[1:], full[1:]))   # (epoch 0 carries first-touch costs on both sides)
    assert sum(dflt[1:]) <= 1.05 * sum(full[1:]), (kind, sum(dflt[1:]), sum(full[1:]))
    assert worst <= 1.12, (kind, worst, dflt, full)
