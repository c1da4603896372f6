"""precision='exact' (csrc/bmu_exact.hpp): the BMUs of the float32 parity kernel, bit for bit, through the IEEE-half MFMA
screen + float32 re-score.  The checker here is the float32 kernel itself (whose own parity with the reference is
pinned by tests/test_gpu_parity.py): every test demands IDENTICAL ids, never "near".  GPU only (`-m gpu`)."""
import numpy as np
import pytest

from oracle import som_oracle as O

pytestmark = pytest.mark.gpu
F32 = np.float32


def engine(X, Y, D, **kw):
    from xpysom_dask_amd.engine import HipEngine
    return HipEngine(X, Y, D, **kw)


def both(X, Y, D, w, data, sigma=3.0, eta=0.4):
    """(ids from the epoch path, ids from the query path) of 'f32' and of 'exact' + the exact engine's stats."""
    out = {}
    for p in ("f32", "exact"):
        e = engine(X, Y, D, precision=p)
        e.set_weights(w)
        e.set_data(data)
        e.epoch_accumulate(sigma, eta, True)
        num, den, bmu = e.epoch_fetch()
        q = e.bmu(data[:777])
        out[p] = (bmu, q, num, den, e.exact_stats() if p == "exact" else None)
        e.close()
    return out


SHAPES = [(6, 6, 4, 150), (20, 24, 128, 5000), (30, 30, 17, 3001), (64, 64, 32, 20000), (40, 50, 100, 4000),
          (128, 128, 64, 9000), (3, 5, 1, 200), (37, 1, 9, 500), (1, 1, 5, 64), (25, 25, 127, 1000), (10, 7, 33, 129)]


@pytest.mark.parametrize("X,Y,D,n", SHAPES)
def test_exact_bmus_are_the_float32_bmus_on_a_seeded_codebook(X, Y, D, n):
    data = O.gaussian_blobs(n, D, seed=D + n)
    w = O.default_codebook(X, Y, D, 11).astype(F32)
    r = both(X, Y, D, w, data)
    assert np.array_equal(r["exact"][0], r["f32"][0])
    assert np.array_equal(r["exact"][1], r["f32"][1])
    # the update path is the shared float32 one: identical BMUs give identical accumulators
    assert np.array_equal(r["exact"][2], r["f32"][2]) and np.array_equal(r["exact"][3], r["f32"][3])
    rows, fb, _ = r["exact"][4]
    assert rows >= n and fb <= max(2, n // 100)          # the float32 fallback kernel is the exception, not the path


@pytest.mark.parametrize("X,Y,D,n,T", [(24, 24, 16, 6000, 8), (64, 64, 32, 20000, 6), (48, 40, 128, 8000, 5)])
def test_exact_training_is_bitwise_the_float32_training(X, Y, D, n, T):
    """Smooth maps (large sigma: neighbouring units nearly identical, hundreds of near-ties per row) are where a
    screen is most often unsure.  Identical BMUs in every epoch <=> identical codebooks after T epochs."""
    from xpysom_dask_amd import XPySom
    data = O.gaussian_blobs(n, D, seed=5)
    ws = {}
    for p in ("f32", "exact"):
        som = XPySom(X, Y, D, random_seed=3, precision=p)
        som.train(data, T)
        ws[p] = som._weights.copy()
        if p == "exact":
            rows, fb, _ = som._engine().exact_stats()
            assert rows == n * T and fb <= rows // 50
    assert np.array_equal(ws["exact"], ws["f32"])


def test_exact_on_a_trained_smooth_map_state_by_state():
    """Epoch by epoch from the float32 trajectory's own codebooks: the ids of every state agree."""
    X, Y, D, n, T = 64, 64, 32, 16384, 6
    data = O.gaussian_blobs(n, D, seed=9)
    w = O.default_codebook(X, Y, D, 2).astype(F32)
    f = engine(X, Y, D, precision="f32")
    x = engine(X, Y, D, precision="exact")
    f.set_data(data)
    x.set_data(data)
    f.set_weights(w)
    for t in range(T):
        sig, eta = O.exponential_decay(32.0, 1.0, t, T), O.exponential_decay(0.5, 0.01, t, T)
        x.set_weights(f.get_weights())
        f.epoch_accumulate(sig, eta, True)
        x.epoch_accumulate(sig, eta, True)
        bf, bx = f.epoch_fetch()[2], x.epoch_fetch()[2]
        assert np.array_equal(bf, bx), "epoch %d: %d rows differ" % (t, (bf != bx).sum())
        f.epoch_merge()
    rows, fb, _ = x.exact_stats()
    assert fb <= rows // 50
    f.close()
    x.close()


def test_exact_ties_and_degenerate_rows():
    """Exact ties (duplicated units, an all-zero codebook, zero rows) resolve to the lowest raveled id as numpy.argmin
    does; small-integer data makes every product exact, so the answer is known without any kernel."""
    rng = np.random.RandomState(0)
    X, Y, D = 9, 8, 8
    w = rng.randint(-3, 4, size=(X * Y, D)).astype(F32)
    w[40] = w[7]
    w[55] = w[7]
    w[3] = 0
    data = rng.randint(-3, 4, size=(500, D)).astype(F32)
    data[10] = 0
    data[11] = w[7]
    want = (-2 * data.astype(np.float64) @ w.T.astype(np.float64) + (w.astype(np.float64) ** 2).sum(1)[None]).argmin(1)
    r = both(X, Y, D, w.reshape(X, Y, D), data)
    assert np.array_equal(r["f32"][0], want)
    assert np.array_equal(r["exact"][0], want)
    zero = both(X, Y, D, np.zeros((X, Y, D), F32), data)
    assert (zero["exact"][0] == 0).all() and (zero["f32"][0] == 0).all()


def test_exact_near_ties_below_the_screen_resolution():
    """Units that differ from each other by a few float32 ulps: far below what the half-precision screen resolves, so the
    re-score decides every row, and it must decide as the float32 kernel does."""
    rng = np.random.RandomState(4)
    X, Y, D, n = 16, 16, 64, 3000
    base = rng.randn(1, D).astype(F32)
    w = np.repeat(base, X * Y, axis=0)
    w *= (1.0 + rng.randint(-4, 5, size=(X * Y, 1)) * 2.0 ** -22).astype(F32)      # clusters of near-identical units
    data = (base + 0.05 * rng.randn(n, D)).astype(F32)
    r = both(X, Y, D, w.reshape(X, Y, D), data)
    assert np.array_equal(r["exact"][0], r["f32"][0])
    assert np.array_equal(r["exact"][1], r["f32"][1])


def test_exact_rows_with_nan_and_inf_take_the_float32_fallback():
    X, Y, D, n = 12, 12, 24, 900
    data = O.gaussian_blobs(n, D, seed=1)
    data[5, 3] = np.nan
    data[77] = np.inf
    data[200, 0] = -np.inf
    data[300] = 1e30                                      # finite, but its products overflow float32
    w = O.default_codebook(X, Y, D, 5).astype(F32)
    r = both(X, Y, D, w, data)
    assert np.array_equal(r["exact"][0], r["f32"][0])
    rows, fb, _ = r["exact"][4]
    assert fb >= 3


def test_exact_many_candidate_groups_overflow_to_the_fallback():
    """A codebook of identical units: every group is a candidate of every row (more than the list holds)."""
    X, Y, D, n = 128, 64, 16, 700
    w = np.ones((X, Y, D), F32)
    data = O.gaussian_blobs(n, D, seed=8)
    r = both(X, Y, D, w, data)
    assert (r["f32"][0] == 0).all() and np.array_equal(r["exact"][0], r["f32"][0])
    rows, fb, _ = r["exact"][4]
    assert fb >= n                                        # 128 groups > 64 list entries: all rows through the float32 kernel


def test_exact_magnitudes_and_streamed_chunks():
    X, Y, D, n = 20, 20, 48, 4096
    for scale in (1e-3, 1.0, 1e3):
        data = (O.gaussian_blobs(n, D, seed=2) * scale).astype(F32)
        w = (O.default_codebook(X, Y, D, 6) * scale).astype(F32)
        r = both(X, Y, D, w, data)
        assert np.array_equal(r["exact"][0], r["f32"][0]), scale
    # streamed chunks (pageable): same sums as the resident epoch of the float32 engine
    data = O.gaussian_blobs(n, D, seed=2)
    w = O.default_codebook(X, Y, D, 6).astype(F32)
    outs = {}
    for p in ("f32", "exact"):
        e = engine(X, Y, D, precision=p)
        e.set_weights(w)
        e.stream_epoch_accumulate([data[:1000], data[1000:1001], data[1001:]], 2.0, 0.3, True)
        outs[p] = e.epoch_fetch(want_bmu=False)[:2]
        e.close()
    assert np.array_equal(outs["exact"][0], outs["f32"][0]) and np.array_equal(outs["exact"][1], outs["f32"][1])


def test_exact_falls_back_to_float32_kernels_where_the_screen_does_not_apply():
    """input_len > 128 and the non-euclidean distances: 'exact' is served by the float32 kernels themselves."""
    X, Y, n = 12, 12, 600
    for D, dist in ((200, "euclidean"), (900, "euclidean"), (32, "cosine"), (16, "manhattan")):
        data = np.abs(O.gaussian_blobs(n, D, seed=3))
        w = np.abs(O.default_codebook(X, Y, D, 4)).astype(F32)
        ids = {}
        for p in ("f32", "exact"):
            e = engine(X, Y, D, precision=p, distance=dist)
            e.set_weights(w)
            ids[p] = e.bmu(data)
            e.close()
        assert np.array_equal(ids["exact"], ids["f32"])


def test_exact_class_surface():
    from xpysom_dask_amd import XPySom
    data = O.gaussian_blobs(500, 6, seed=1)
    a = XPySom(7, 7, 6, random_seed=1, precision="exact").train(data, 3)
    b = XPySom(7, 7, 6, random_seed=1, precision="f32").train(data, 3)
    assert np.array_equal(a._weights, b._weights)
    assert a.winner(data) == b.winner(data)
    assert a.quantization_error(data) == b.quantization_error(data)
    assert a.topographic_error(data) == b.topographic_error(data)


def test_the_mfma_rounding_the_bound_charges_for_is_measured():
    """exact_bound() charges KAPPA = 6 ulps (of the largest magnitude among accumulator, result and sum of |products|) per
    v_mfma_f32_16x16x32: the hardware's internal summation order and width are not documented, so the number is
    MEASURED here (som_debug_mfma16: one MFMA on given operands against float64) over operand scales, accumulators
    from 0 to 2^37, cancelling and non-cancelling products, IEEE half and bfloat16.  Observed <= 2.4; fail above 3."""
    e = engine(4, 4, 4, precision="f32")
    rs = np.random.RandomState(0)
    worst = 0.0
    for trial in range(1500):
        f16 = trial % 2 == 0
        a = rs.randn(16, 32) * 2.0 ** rs.randint(-6, 12)
        b = rs.randn(32, 16) * 2.0 ** rs.randint(-6, 12)
        if trial % 5 == 1:
            a[:, 2:] = 0
        if trial % 5 == 2:
            a, b = np.abs(a), np.abs(b)
        if f16:
            a16, b16 = a.astype(np.float16), b.astype(np.float16)
            av, bv = a16.astype(np.float64), b16.astype(np.float64)
        else:                                             # bfloat16: the upper half of the float32 pattern (truncated)
            a16 = (a.astype(F32).view(np.uint32) >> 16).astype(np.uint16)
            b16 = (b.astype(F32).view(np.uint32) >> 16).astype(np.uint16)
            av = (a16.astype(np.uint32) << 16).view(F32).astype(np.float64)
            bv = (b16.astype(np.uint32) << 16).view(F32).astype(np.float64)
        cs = [0.0, 1.0, 2.0 ** 10, 2.0 ** 20, 2.0 ** 30, 2.0 ** 37][rs.randint(0, 6)]
        c = ((rs.rand(16, 16) + 0.5) * cs * (1 if trial % 7 else -1)).astype(F32)
        d = e.debug_mfma16(a16, b16, c, f16=f16)
        ex = av @ bv + c.astype(np.float64)
        mag = np.maximum(np.maximum(np.abs(c.astype(np.float64)), np.abs(ex)), np.abs(av) @ np.abs(bv))
        ulp = 2.0 ** (np.floor(np.log2(np.maximum(mag, 1e-300))) - 23)
        worst = max(worst, float((np.abs(d.astype(np.float64) - ex) / ulp).max()))
    e.close()
    assert worst <= 3.0, worst


def test_exact_in_several_screen_passes(monkeypatch):
    """Large row sets are screened in passes (the group-minimum matrix of a pass stays within 4 GiB: 1 Mi rows at
    256 x 256); SOM_EXACT_PASS_ROWS forces passes of 1 024 / 2 048 rows so that the pass loop -- slices of the rows, of the
    merge keys, of the norms; a last short pass; fallback rows in a later pass -- runs on test-sized data."""
    X, Y, D, n = 40, 40, 64, 7001
    data = O.gaussian_blobs(n, D, seed=12)
    data[5000] = np.nan                                   # a fallback row in the fifth / third pass
    w = O.smooth_sheet_codebook(X, Y, D, 5, amplitude=0.3, centre=np.nanmean(data.astype(np.float64), axis=0))
    want = both(X, Y, D, w, data)["f32"][0]
    for rows in ("1024", "2048"):
        monkeypatch.setenv("SOM_EXACT_PASS_ROWS", rows)
        e = engine(X, Y, D, precision="exact")
        e.set_weights(w)
        e.set_data(data)
        e.epoch_accumulate(3.0, 0.4, True)
        got = e.epoch_fetch()[2]
        r, fb, passes = e.exact_stats()
        assert passes == -(-n // int(rows)) and r == n and fb >= 1
        assert np.array_equal(got, want), (rows, int((got != want).sum()))
        assert np.array_equal(e.bmu(data[:3000]), want[:3000])
        e.close()


# ---- beyond 128 features: the wide screen (maps of >= 4096 units, euclidean and cosine) -------------------------------

def both_dist(X, Y, D, w, data, dist, sigma=3.0, eta=0.4):
    out = {}
    for p in ("f32", "exact"):
        e = engine(X, Y, D, precision=p, distance=dist)
        e.set_weights(w)
        e.set_data(data)
        e.epoch_accumulate(sigma, eta, True)
        num, den, bmu = e.epoch_fetch()
        q = e.bmu(data[:777])
        out[p] = (bmu, q, num, den, e.exact_stats() if p == "exact" else None)
        e.close()
    return out


WIDE_SHAPES = [(64, 64, 200, 3000), (64, 64, 784, 2500), (80, 60, 129, 1111), (64, 70, 300, 4097), (100, 100, 517, 2000),
               (64, 64, 800, 700), (128, 64, 160, 5000)]


@pytest.mark.parametrize("dist", ["euclidean", "cosine"])
@pytest.mark.parametrize("X,Y,D,n", WIDE_SHAPES)
def test_exact_wide_bmus_are_the_float32_bmus(X, Y, D, n, dist):
    data = O.gaussian_blobs(n, D, seed=D + n)
    w = O.default_codebook(X, Y, D, 11).astype(F32)
    r = both_dist(X, Y, D, w, data, dist)
    assert np.array_equal(r["exact"][0], r["f32"][0])
    assert np.array_equal(r["exact"][1], r["f32"][1])
    assert np.array_equal(r["exact"][2], r["f32"][2]) and np.array_equal(r["exact"][3], r["f32"][3])
    rows, fb, _ = r["exact"][4]
    assert rows >= n and fb <= max(2, n // 100)          # the screen serves: the float32 fallback is the exception


@pytest.mark.parametrize("dist", ["euclidean", "cosine"])
def test_exact_wide_training_is_bitwise_the_float32_training(dist):
    from xpysom_dask_amd import XPySom
    X, Y, D, n, T = 64, 64, 200, 6000, 5
    data = np.abs(O.gaussian_blobs(n, D, seed=5)) + F32(0.05)
    ws = {}
    for p in ("f32", "exact"):
        som = XPySom(X, Y, D, random_seed=3, precision=p, activation_distance=dist)
        som.train(data, T)
        ws[p] = som._weights.copy()
        if p == "exact":
            rows, fb, _ = som._engine().exact_stats()
            assert rows == n * T and fb <= rows // 50
    assert np.array_equal(ws["exact"], ws["f32"])


@pytest.mark.parametrize("dist", ["euclidean", "cosine"])
def test_exact_wide_ties_degenerate_rows_and_magnitudes(dist):
    X, Y, D, n = 64, 64, 256, 2048
    rng = np.random.default_rng(7)
    w = rng.standard_normal((X * Y, D)).astype(F32)
    w[100] = w[7]; w[4000] = w[7]; w[3000:3010] = w[2999]              # duplicate units: the first one wins
    data = rng.standard_normal((n, D)).astype(F32)
    data[0] = w[7]; data[1] = w[4000] * F32(1.0000001); data[2] = 0.0   # on a unit; next to it; a zero row
    data[3] = data[4]                                                   # equal rows
    data[5] = np.nan; data[6, 3] = np.inf; data[7] = 1e-30; data[8] = 1e18
    data[100:200] *= F32(1e-3); data[200:300] *= F32(1e3)               # a wide spread of row norms
    data[300:400] = w[rng.integers(0, X * Y, 100)] + F32(1e-4) * rng.standard_normal((100, D)).astype(F32)
    r = both_dist(X, Y, D, w.reshape(X, Y, D), data, dist)
    assert np.array_equal(r["exact"][0], r["f32"][0])
    assert np.array_equal(r["exact"][1], r["f32"][1])
    for scale in (1e-12, 1e9):                                          # the whole problem far from 1
        r = both_dist(X, Y, D, (w * F32(scale)).reshape(X, Y, D), data * F32(scale), dist)
        assert np.array_equal(r["exact"][0], r["f32"][0])


def test_exact_wide_cosine_with_unusable_units_falls_back_whole():
    """a zero unit (the float32 kernel's 0/0 -> nan_to_num rule) and units outside the window in which |x|^2 |w|^2 stays a
    normal float32: the bound does not model them, every row goes to the float32 kernel."""
    X, Y, D, n = 64, 64, 130, 900
    rng = np.random.default_rng(3)
    for bad in (0.0, 1e-25, 1e25):
        w = rng.standard_normal((X * Y, D)).astype(F32)
        w[17] = w[17] * F32(bad)
        data = rng.standard_normal((n, D)).astype(F32)
        r = both_dist(X, Y, D, w.reshape(X, Y, D), data, "cosine")
        assert np.array_equal(r["exact"][0], r["f32"][0])
        rows, fb, _ = r["exact"][4]
        assert fb == rows


def test_exact_wide_on_a_smooth_sheet():
    """neighbouring units nearly identical: hundreds of near-ties per row."""
    X, Y, D, n = 64, 64, 784, 3000
    w = O.smooth_sheet_codebook(X, Y, D, seed=2).astype(F32)
    data = O.gaussian_blobs(n, D, seed=9)
    for dist in ("euclidean", "cosine"):
        r = both_dist(X, Y, D, w, data, dist)
        assert np.array_equal(r["exact"][0], r["f32"][0])


@pytest.mark.parametrize("two_round", ["0", "1"])
@pytest.mark.parametrize("X,Y,D,dist", [(64, 64, 32, "euclidean"), (64, 64, 200, "cosine")])
def test_exact_one_round_and_two_round_rescore_agree_with_float32(monkeypatch, two_round, X, Y, D, dist):
    """SOM_EXACT_TWO_ROUND forces either re-score scheme on either tiling (the default is one round up to 128 features, two
    beyond): both must return the float32 kernel's BMUs, on a random and on a smooth codebook."""
    monkeypatch.setenv("SOM_EXACT_TWO_ROUND", two_round)
    n = 3000
    data = O.gaussian_blobs(n, D, seed=21)
    for w in (O.default_codebook(X, Y, D, 5).astype(F32), O.smooth_sheet_codebook(X, Y, D, seed=4).astype(F32)):
        r = both_dist(X, Y, D, w, data, dist)
        assert np.array_equal(r["exact"][0], r["f32"][0])
        assert np.array_equal(r["exact"][1], r["f32"][1])
        rows, fb, _ = r["exact"][4]
        assert rows >= n and fb <= n // 20


@pytest.mark.parametrize("seed_on", ["1", "0"])
def test_exact_seeded_epochs_return_the_float32_bmus(monkeypatch, seed_on):
    """From the second epoch on resident rows the screen and the select kernel are capped by a seed computed from last epoch's
    BMUs (exact_seed_kernel).  Any ids give a valid cap: after ordinary merges, after a codebook that moved every unit
    somewhere else, and after set_data with other rows, every epoch's BMUs are the float32 kernel's."""
    monkeypatch.setenv("SOM_EXACT_SEED", seed_on)
    X, Y, D, n = 48, 48, 64, 9000
    data = O.gaussian_blobs(n, D, seed=3)
    other = O.gaussian_blobs(n, D, seed=4)[::-1].copy()
    w = O.smooth_sheet_codebook(X, Y, D, seed=6).astype(F32)
    eng = {p: engine(X, Y, D, precision=p) for p in ("f32", "exact")}
    for e in eng.values():
        e.set_weights(w)
        e.set_data(data)
    def step(sigma, eta):
        ids = {}
        for p, e in eng.items():
            e.epoch_accumulate(sigma, eta, True)
            ids[p] = e.epoch_fetch()[2]
        assert np.array_equal(ids["exact"], ids["f32"])
    for sigma in (12.0, 8.0, 4.0):                                     # ordinary epochs: merge, next epoch
        step(sigma, 0.4)
        for e in eng.values():
            e.epoch_merge()
    for e in eng.values():                                             # every unit somewhere else
        e.set_weights(w[::-1, ::-1].copy() * F32(1.7))
    step(6.0, 0.3)
    for e in eng.values():                                             # other rows: the old ids mean nothing
        e.set_data(other)
    step(6.0, 0.3)
    step(3.0, 0.3)
    rows, fb, _ = eng["exact"].exact_stats()
    assert rows == 6 * n and fb <= rows // 50
    for e in eng.values():
        e.close()


def test_exact_falls_back_to_smaller_passes_when_the_scratch_is_refused(monkeypatch):
    """A device that cannot give a pass its scratch (here: the test hook refuses anything above 2 500 rows) gets passes of half
    the rows, and half again, instead of an error; the BMUs are the float32 kernel's either way."""
    monkeypatch.setenv("SOM_EXACT_DEBUG_REFUSE_ABOVE", "2500")
    X, Y, D, n = 32, 32, 24, 9000
    data = O.gaussian_blobs(n, D, seed=12)
    w = O.default_codebook(X, Y, D, 3).astype(F32)
    r = both(X, Y, D, w, data)
    assert np.array_equal(r["exact"][0], r["f32"][0]) and np.array_equal(r["exact"][1], r["f32"][1])
    rows, fb, passes = r["exact"][4]
    assert passes >= 4 + 1                                   # 9 000 rows in passes of 2 304, + the 777-row query


@pytest.mark.parametrize("X,Y,D,n", [(512, 512, 64, 20000), (1024, 256, 17, 9000)])
def test_exact_on_maps_of_a_quarter_million_units(X, Y, D, n):
    """4 096 groups on the resident screen (configs[4]'s map size with short rows): the per-group lists, the tile table and the
    stored-row masks at their largest; two epochs (the second one seeded), random and smooth codebook."""
    data = O.gaussian_blobs(n, D, seed=5)
    for w in (O.default_codebook(X, Y, D, 3).astype(F32), O.smooth_sheet_codebook(X, Y, D, seed=2).astype(F32)):
        ids = {}
        for p in ("f32", "exact"):
            e = engine(X, Y, D, precision=p)
            e.set_weights(w)
            e.set_data(data)
            e.epoch_accumulate(20.0, 0.4, True)
            b1 = e.epoch_fetch()[2]
            e.epoch_merge()
            e.epoch_accumulate(10.0, 0.3, True)
            ids[p] = (b1, e.epoch_fetch()[2])
            e.close()
        assert np.array_equal(ids["exact"][0], ids["f32"][0]) and np.array_equal(ids["exact"][1], ids["f32"][1])


# ----------------------------------------------------------------------------- patch order
@pytest.mark.parametrize("X,Y", [(16, 24), (32, 32), (64, 8), (13, 21), (9, 100), (70, 3)])
def test_exact_patch_order_ties_prefer_the_lowest_unit(monkeypatch, X, Y):
    """The exact mode's operand images hold the units patch by patch (8 x 8 units of the map per 64-unit group where the
    sides are multiples of 8; bands of 8 map rows cut every 64 units otherwise: som_common.hpp); the answer must not
    know: among equal scores the lowest UNIT id wins,
    within a patch, across patches, and for rows the float32 fallback settles (several passes here, so that the float32
    image changes order back and forth between them).  Small integers: every product exact, numpy float64 is the checker."""
    monkeypatch.setenv("SOM_EXACT_PASS_ROWS", "1024")
    rng = np.random.RandomState(2)
    D, n = 8, 3000
    proto = rng.randint(-3, 4, size=(5, D)).astype(F32)
    w = proto[rng.randint(0, 5, size=X * Y)]                              # five distinct units, scattered over all patches
    data = rng.randint(-3, 4, size=(n, D)).astype(F32)
    data[::7] = proto[rng.randint(0, 5, size=len(data[::7]))]             # rows ON units: a whole class of exact ties
    want = (-2 * data.astype(np.float64) @ w.T.astype(np.float64) + (w.astype(np.float64) ** 2).sum(1)[None]).argmin(1)
    data[100, 2] = np.nan; data[1500] = np.inf; data[2100, 1] = np.nan    # fallback rows in passes 0, 1 and 2
    ok = np.ones(n, bool); ok[[100, 1500, 2100]] = False
    r = both(X, Y, D, w.reshape(X, Y, D), data)
    assert np.array_equal(r["f32"][0][ok], want[ok])
    assert np.array_equal(r["exact"][0], r["f32"][0])
    assert np.array_equal(r["exact"][1], r["f32"][1])
    rows, fb, passes = r["exact"][4]
    assert passes >= 4 and fb >= 3


@pytest.mark.parametrize("X,Y,D,n,dist", [(64, 64, 32, 12000, "euclidean"), (16, 8, 128, 3000, "euclidean"),
                                           (72, 64, 200, 3000, "cosine"), (64, 72, 133, 3000, "euclidean"),
                                           (100, 100, 64, 8000, "euclidean"), (90, 75, 200, 3000, "cosine"),
                                           (67, 131, 17, 5000, "euclidean")])
def test_exact_patch_order_equals_strip_order(monkeypatch, X, Y, D, n, dist):
    """SOM_EXACT_PATCH=0 keeps the units' own order (a group = 64 consecutive units): same BMUs, same accumulators, on a
    smooth map (where the patches save candidate groups) and in the analysis calls that run float32 kernels on the same
    handle in between (top-2, the distance matrix: they want the float32 image in the units' own order)."""
    from xpysom_dask_amd import XPySom
    data = O.gaussian_blobs(n, D, seed=3)
    if dist == "cosine":
        data = np.abs(data)
    som = XPySom(X, Y, D, sigma=min(X, Y) / 2.0, random_seed=1, precision="f32", activation_distance=dist)
    som.train(data, 2)                                                    # a smooth early map
    w = som._weights.astype(F32)
    ref = engine(X, Y, D, precision="f32", distance=dist)
    ref.set_weights(w); ref.set_data(data)
    ref.epoch_accumulate(4.0, 0.3, True)
    rnum, rden, rbmu = ref.epoch_fetch()
    r1, r2 = ref.bmu_top2(data[:500])
    counts = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("SOM_EXACT_PATCH", mode)
        e = engine(X, Y, D, precision="exact", distance=dist)
        e.set_weights(w); e.set_data(data)
        for _ in range(2):                                                # (the second epoch: seeded from the first one's BMUs)
            e.epoch_accumulate(4.0, 0.3, True)
            num, den, bmu = e.epoch_fetch()
            assert np.array_equal(bmu, rbmu) and np.array_equal(num, rnum) and np.array_equal(den, rden)
            t1, t2 = e.bmu_top2(data[:500])
            assert np.array_equal(t1, r1) and np.array_equal(t2, r2)
            assert np.array_equal(e.bmu(data[:700]), rbmu[:700])
        counts[mode] = e.exact_last_counts(min(n, 700)).mean()
        rows, fb, _ = e.exact_stats()
        assert fb <= rows // 50
        e.close()
    ref.close()
    if X >= 64:
        assert counts["1"] < counts["0"]                                  # fewer candidate groups per row: the point of it


@pytest.mark.parametrize("X,Y", [(16, 24), (21, 19)])
def test_exact_patch_order_copy_stays_in_step_with_the_merge(X, Y):
    """The merge writes the codebook and, in the same launch, its patch-order copy -- only where a unit's denominator
    is not zero, and only if the copy was in step before (after a forced-BMU epoch on a fresh upload it is not: the
    next BMU search permutes again).  A tiny sigma leaves most units untouched; every epoch's BMUs equal float32's."""
    D, n = 12, 4000
    data = O.gaussian_blobs(n, D, seed=8)
    w = O.default_codebook(X, Y, D, 3).astype(F32)
    es = {p: engine(X, Y, D, precision=p) for p in ("f32", "exact")}
    forced = np.random.RandomState(0).randint(0, X * Y, size=n).astype(np.int32)
    trace = {}
    for p, e in es.items():
        e.set_weights(w); e.set_data(data)
        e.epoch_accumulate_forced(forced, 2.0, 0.5, True)                 # no BMU search yet: the copy was never made
        e.epoch_merge()
        out = []
        for sig in (3.0, 0.3, 0.3, 1.5, 0.2):
            e.epoch_accumulate(sig, 0.5, True)
            out.append(e.epoch_fetch()[2].copy())
            e.epoch_merge()
        out.append(e.get_weights().copy())
        trace[p] = out
        e.close()
    for a, b in zip(trace["f32"], trace["exact"]):
        assert np.array_equal(a, b)


# ----------------------------------------------------------------------------- block skipping (csrc/exact_skip.hpp)
@pytest.mark.parametrize("X,Y,D,n,pass_rows", [(64, 64, 32, 20000, 0), (40, 50, 100, 6000, 0), (33, 17, 128, 5000, 1024),
                                               (128, 128, 16, 30000, 0), (16, 16, 7, 3000, 1024), (100, 100, 64, 9000, 4096)])
def test_exact_block_skipping_trains_the_float32_map(monkeypatch, X, Y, D, n, pass_rows):
    """From the second epoch on the screen runs, per 256-row tile of rows sorted by their last BMU's patch, only the groups
    whose centroid-and-radius bound leaves some row a chance (SOM_EXACT_SKIP=2: on every map of >= 2 groups).  Same BMUs in
    every epoch of a whole schedule -- smooth maps that skip nothing, trained maps that skip most -- hence the same
    codebook, bit for bit; and on the trained map most blocks were indeed not run."""
    monkeypatch.setenv("SOM_EXACT_SKIP", "2")
    if pass_rows:
        monkeypatch.setenv("SOM_EXACT_PASS_ROWS", str(pass_rows))
    data = O.gaussian_blobs(n, D, seed=4)
    w = O.default_codebook(X, Y, D, 6).astype(F32)
    f = engine(X, Y, D, precision="f32")
    x = engine(X, Y, D, precision="exact")
    for e in (f, x):
        e.set_weights(w); e.set_data(data)
    T = 8
    for t in range(T):
        sig, eta = O.exponential_decay(min(X, Y) / 2.0, 1.0, t, T), O.exponential_decay(0.5, 0.01, t, T)
        f.epoch_accumulate(sig, eta, True)
        x.epoch_accumulate(sig, eta, True)
        bf, bx = f.epoch_fetch()[2], x.epoch_fetch()[2]
        assert np.array_equal(bf, bx), "epoch %d: %d rows differ" % (t, (bf != bx).sum())
        f.epoch_merge(); x.epoch_merge()
    assert np.array_equal(f.get_weights(), x.get_weights())
    run, total = x.exact_skip_stats()
    rows, fb, _ = x.exact_stats()
    assert rows == n * T and fb <= rows // 50
    assert 0 < run <= total                               # (a map of a few groups: every tile may need them all)
    f.close(); x.close()


def test_exact_block_skipping_skips_most_of_a_trained_benchmark_map(monkeypatch):
    """configs[2]'s map and data (256 x 256 x 128, Gaussian blobs), 32 768 rows, the benchmark's schedule: from the third
    epoch on a tenth to a quarter of the blocks run (tools/skip_probe.py counted 7-28 % in real arithmetic).  (Mode 2: a
    plan in every epoch; the default mode pauses the plan for two epochs after two that kept > 97 % of the blocks.)"""
    monkeypatch.setenv("SOM_EXACT_SKIP", "2")
    from xpysom_dask_amd.synthetic import gaussian_blobs
    X = Y = 256; D = 128; n = 32768; T = 10
    data = gaussian_blobs(n, D)
    rs = np.random.RandomState(1234)
    w = rs.rand(X, Y, D) * 2 - 1
    w = (w / np.linalg.norm(w, axis=-1, keepdims=True)).astype(F32)
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
        assert np.array_equal(f.epoch_fetch()[2], x.epoch_fetch()[2]), t
        f.epoch_merge(); x.epoch_merge()
    # (epoch 0: a random codebook -- the scout's plan runs and keeps (nearly) everything)
    assert shares[0] > 0.9 and max(shares[3:]) < 0.5 and min(shares[3:]) < 0.2, shares
    f.close(); x.close()


def test_exact_block_skipping_with_ties_fallback_rows_and_a_moved_codebook(monkeypatch):
    """What the bound must survive: exact ties (duplicated units in different patches), NaN / infinite / zero rows, a
    codebook replaced between epochs (the last BMUs then say nothing: a valid but useless bound), queries in between."""
    monkeypatch.setenv("SOM_EXACT_SKIP", "2")
    rng = np.random.RandomState(5)
    X, Y, D, n = 24, 32, 8, 4000
    proto = rng.randint(-3, 4, size=(7, D)).astype(F32)
    w = proto[rng.randint(0, 7, size=X * Y)].reshape(X, Y, D)
    data = rng.randint(-3, 4, size=(n, D)).astype(F32)
    data[::5] = proto[rng.randint(0, 7, size=len(data[::5]))]
    data[7] = 0.0; data[100, 2] = np.nan; data[1500] = np.inf; data[2100] = 1e30
    f = engine(X, Y, D, precision="f32"); x = engine(X, Y, D, precision="exact")
    for e in (f, x):
        e.set_weights(w); e.set_data(data)
    w2 = (w[::-1, ::-1] * F32(2.0)).copy()
    for t, sig in enumerate((4.0, 4.0, 2.0, 1.0, 1.0)):
        if t == 3:
            f.set_weights(w2); x.set_weights(w2)
        f.epoch_accumulate(sig, 0.3, True); x.epoch_accumulate(sig, 0.3, True)
        assert np.array_equal(f.epoch_fetch()[2], x.epoch_fetch()[2]), t
        assert np.array_equal(f.bmu(data[:300]), x.bmu(data[:300]))
        if t != 1:
            f.epoch_merge(); x.epoch_merge()
    f.close(); x.close()


def test_exact_block_skipping_off_and_auto(monkeypatch):
    """SOM_EXACT_SKIP=0: every block runs.  Default: on from 4 096 units."""
    from xpysom_dask_amd.synthetic import gaussian_blobs
    X, Y, D, n = 64, 64, 64, 8000
    data = gaussian_blobs(n, D, seed=2, centres=16, spread=6.0)            # well separated: a trained map skips a lot
    w = O.default_codebook(X, Y, D, 1).astype(F32)
    got = {}
    for mode in ("0", None):
        if mode is None:
            monkeypatch.delenv("SOM_EXACT_SKIP", raising=False)
        else:
            monkeypatch.setenv("SOM_EXACT_SKIP", mode)
        e = engine(X, Y, D, precision="exact")
        e.set_weights(w); e.set_data(data)
        for sig in (8.0, 2.0, 1.0, 1.0):
            e.epoch(sig, 0.3, True)
        e.epoch_accumulate(1.0, 0.3, True)
        got[mode] = (e.epoch_fetch()[2], e.get_weights(), e.exact_skip_stats())
        e.close()
    assert np.array_equal(got["0"][0], got[None][0]) and np.array_equal(got["0"][1], got[None][1])
    assert got["0"][2][0] == got["0"][2][1] and got[None][2][0] < got[None][2][1]


def test_exact_block_skipping_with_nan_units_in_the_codebook(monkeypatch):
    """A NaN row poisons the units its neighbourhood reaches at the merge; rows whose last BMU is such a unit have no bound
    (found by the fuzzer: an fmax had swallowed that NaN and bounded those rows by zero) and must need every group."""
    monkeypatch.setenv("SOM_EXACT_SKIP", "2")
    X, Y, D, n = 16, 80, 32, 5000
    data = O.gaussian_blobs(n, D, seed=11)
    data[1234] = np.nan
    w = (data[:1].mean(0) + 1e-3 * np.random.RandomState(0).randn(X, Y, D)).astype(F32) * F32(100.0)   # epoch 1: one BMU for all
    f = engine(X, Y, D, precision="f32"); x = engine(X, Y, D, precision="exact")
    for e in (f, x):
        e.set_weights(w); e.set_data(data)
    for sig in (1.0, 0.7, 0.7):
        f.epoch_accumulate(sig, 0.5, True); x.epoch_accumulate(sig, 0.5, True)
        assert np.array_equal(f.epoch_fetch()[2], x.epoch_fetch()[2])
        f.epoch_merge(); x.epoch_merge()
    assert np.isnan(f.get_weights()).any()
    f.close(); x.close()


def test_exact_block_skipping_without_memory_for_it_runs_every_block(monkeypatch):
    """The sorted pass needs buffers of its own; a device that refuses them (SOM_EXACT_DEBUG_REFUSE_SKIP: test hook) gets
    the full scan -- same ids, no error."""
    monkeypatch.setenv("SOM_EXACT_SKIP", "2")
    monkeypatch.setenv("SOM_EXACT_DEBUG_REFUSE_SKIP", "1")
    X, Y, D, n = 32, 32, 16, 4000
    data = O.gaussian_blobs(n, D, seed=3)
    w = O.default_codebook(X, Y, D, 2).astype(F32)
    f = engine(X, Y, D, precision="f32"); x = engine(X, Y, D, precision="exact")
    for e in (f, x):
        e.set_weights(w); e.set_data(data)
    for sig in (4.0, 2.0, 1.0):
        f.epoch_accumulate(sig, 0.4, True); x.epoch_accumulate(sig, 0.4, True)
        assert np.array_equal(f.epoch_fetch()[2], x.epoch_fetch()[2])
        f.epoch_merge(); x.epoch_merge()
    run, total = x.exact_skip_stats()
    assert run == total > 0
    f.close(); x.close()


def test_exact_block_skipping_pauses_the_plan_on_rows_without_structure():
    """Default mode: two launches in a row whose plans kept more than 97 % of the blocks are followed by two launches without
    a plan (rows of one tight cluster on a big map: every group is near every row).  Same ids either way."""
    from xpysom_dask_amd.synthetic import gaussian_blobs
    X, Y, D, n = 64, 64, 16, 8192
    data = gaussian_blobs(n, D, seed=9, centres=1, spread=0.1)
    w = O.default_codebook(X, Y, D, 3).astype(F32)
    f = engine(X, Y, D, precision="f32"); x = engine(X, Y, D, precision="exact")
    for e in (f, x):
        e.set_weights(w); e.set_data(data)
    for t in range(6):
        f.epoch_accumulate(20.0, 0.3, True); x.epoch_accumulate(20.0, 0.3, True)
        assert np.array_equal(f.epoch_fetch()[2], x.epoch_fetch()[2]), t
        f.epoch_merge(); x.epoch_merge()
    run, total = x.exact_skip_stats()
    assert run > 0.97 * total
    f.close(); x.close()


# ----------------------------------------------------------------------------- the resident sorted pass, the sub-block plan
def _train_states(monkeypatch, env, X, Y, D, n, T, data, w):
    for k in ("SOM_EXACT_RESORT", "SOM_EXACT_SUBBLOCKS", "SOM_EXACT_SKIP", "SOM_EXACT_PASS_ROWS", "SOM_EXACT_SUB44"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    e = engine(X, Y, D, precision="exact")
    e.set_weights(w); e.set_data(data)
    ids = []
    for t in range(T):
        sig, eta = O.exponential_decay(min(X, Y) / 2.0, 1.0, t, T), O.exponential_decay(0.5, 0.01, t, T)
        e.epoch_accumulate(sig, eta, True)
        ids.append(e.epoch_fetch()[2])
        e.epoch_merge()
    out = (ids, e.get_weights(), e.exact_skip_stats(), e.exact_resident_stats())
    e.close()
    return out


def test_exact_resident_order_is_resorted_lazily_and_never_changes_the_ids(monkeypatch):
    """The rows stay resident in the order of their BMU's patch at the time of the last sort; the plan tests every row of a
    tile where it sits, so a stale order costs blocks, never a BMU.  Re-sorting every planned epoch (SOM_EXACT_RESORT=1),
    every third, never again after the first sort (=1000) and the default policy (when the order has gone stale) give the
    same ids in every epoch and the same codebook as precision='f32'."""
    X, Y, D, n, T = 64, 64, 32, 30000, 12
    data = O.gaussian_blobs(n, D, seed=21)
    w = O.default_codebook(X, Y, D, 2).astype(F32)
    f = engine(X, Y, D, precision="f32")
    f.set_weights(w); f.set_data(data)
    ref = []
    for t in range(T):
        f.epoch_accumulate(O.exponential_decay(32.0, 1.0, t, T), O.exponential_decay(0.5, 0.01, t, T), True)
        ref.append(f.epoch_fetch()[2]); f.epoch_merge()
    wf = f.get_weights(); f.close()
    runs = {tag: _train_states(monkeypatch, env, X, Y, D, n, T, data, w)
            for tag, env in (("every", {"SOM_EXACT_RESORT": "1"}), ("third", {"SOM_EXACT_RESORT": "3"}),
                             ("never", {"SOM_EXACT_RESORT": "1000"}), ("default", {"SOM_EXACT_SKIP": "2"}), ("auto", {}))}
    for tag, (ids, wx, skip, res) in runs.items():
        for t in range(T):
            assert np.array_equal(ids[t], ref[t]), (tag, t, int((ids[t] != ref[t]).sum()))
        assert np.array_equal(wx, wf), tag
    planned, sorts = runs["default"][3]                      # (mode 2: a plan in every epoch, the default sort policy)
    assert planned >= T - 3 and 1 <= sorts <= planned, (planned, sorts)     # (a map this small keeps a quarter of its blocks: it sorts often)
    # (default mode: on a map this small a planned launch costs what a full scan costs -- measured -- and the plan is paused)
    assert 1 <= runs["auto"][3][0] <= T
    assert runs["every"][3][1] == runs["every"][3][0] and runs["never"][3][1] == 1
    assert 1 < runs["third"][3][1] < runs["third"][3][0]
    # a fresher order never runs more blocks than the order of the first planned epoch kept forever
    assert runs["every"][2][0] <= runs["never"][2][0]


def test_exact_sub_block_plan_drops_blocks_and_keeps_the_ids(monkeypatch):
    """Level 2 of the plan (the groups' four 16-unit sub-blocks with their own centroids and radii) only removes blocks from
    what level 1 keeps: same ids with it (SOM_EXACT_SKIP=2 forces it on every planned epoch) and without it
    (SOM_EXACT_SUBBLOCKS=0), fewer blocks run with it on a trained map."""
    X, Y, D, n, T = 64, 64, 32, 20000, 8
    data = O.gaussian_blobs(n, D, seed=4)
    w = O.default_codebook(X, Y, D, 6).astype(F32)
    a = _train_states(monkeypatch, {"SOM_EXACT_SKIP": "2"}, X, Y, D, n, T, data, w)
    b = _train_states(monkeypatch, {"SOM_EXACT_SKIP": "2", "SOM_EXACT_SUBBLOCKS": "0"}, X, Y, D, n, T, data, w)
    for t in range(T):
        assert np.array_equal(a[0][t], b[0][t]), t
    assert np.array_equal(a[1], b[1])
    assert a[2][1] == b[2][1] and a[2][0] < 0.9 * b[2][0], (a[2], b[2])
    # the sub-blocks are 4 x 4 squares of the map where both sides are multiples of 8 (som_patch_order); as 2 x 8 strips
    # (SOM_EXACT_SUB44=0: every group's units ascending) the ids are the same
    c = _train_states(monkeypatch, {"SOM_EXACT_SKIP": "2", "SOM_EXACT_SUB44": "0"}, X, Y, D, n, T, data, w)
    for t in range(T):
        assert np.array_equal(a[0][t], c[0][t]), t
    assert np.array_equal(a[1], c[1]) and a[2][1] == c[2][1]


def test_exact_listed_screen_work_queue_equals_one_workgroup_per_tile(monkeypatch):
    """The listed screen as a work queue (the tiles' lists cut into items of about equal length, parts ending at group boundaries,
    a workgroup per slot taking items off one counter) against the grid of one workgroup per tile (SOM_EXACT_QUEUE=0), and with
    items a quarter of the mean list (=25: every tile in several parts): the same ids in every epoch, the same codebook, the same
    blocks run."""
    X, Y, D, n, T = 96, 64, 64, 40000, 8
    data = O.gaussian_blobs(n, D, seed=9)
    w = O.default_codebook(X, Y, D, 3).astype(F32)
    monkeypatch.delenv("SOM_EXACT_QUEUE", raising=False)
    runs = {}
    for tag, q in (("queue", None), ("grid", "0"), ("short", "25"), ("long", "400")):
        env = {"SOM_EXACT_SKIP": "2"}
        if q is not None:
            env["SOM_EXACT_QUEUE"] = q
        runs[tag] = _train_states(monkeypatch, env, X, Y, D, n, T, data, w)
        monkeypatch.delenv("SOM_EXACT_QUEUE", raising=False)
    for tag in ("grid", "short", "long"):
        for t in range(T):
            assert np.array_equal(runs["queue"][0][t], runs[tag][0][t]), (tag, t)
        assert np.array_equal(runs["queue"][1], runs[tag][1]), tag
        assert runs["queue"][2] == runs[tag][2], (tag, runs["queue"][2], runs[tag][2])
    assert runs["queue"][2][0] < 0.5 * runs["queue"][2][1]          # (blocks were skipped: the lists are lists)


def test_exact_new_rows_of_the_same_size_drop_the_resident_order():
    """som_set_data with another row set of the same shape (the allocator may hand back the same address): the sorted copies
    of the old rows must not survive."""
    X, Y, D, n = 64, 64, 16, 12000
    w = O.default_codebook(X, Y, D, 5).astype(F32)
    d1, d2 = O.gaussian_blobs(n, D, seed=1), O.gaussian_blobs(n, D, seed=2)
    f = engine(X, Y, D, precision="f32"); x = engine(X, Y, D, precision="exact")
    for e in (f, x):
        e.set_weights(w)
    for data in (d1, d2, d1):
        for e in (f, x):
            e.set_data(data)
        for sig in (8.0, 3.0, 1.5):
            f.epoch_accumulate(sig, 0.3, True); x.epoch_accumulate(sig, 0.3, True)
            assert np.array_equal(f.epoch_fetch()[2], x.epoch_fetch()[2])
            f.epoch_merge(); x.epoch_merge()
    assert x.exact_resident_stats()[1] >= 3               # at least one sort per row set
    f.close(); x.close()
