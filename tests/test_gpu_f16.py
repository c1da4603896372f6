"""precision='f16': the bf16 kernels on IEEE half operands (csrc/som_common.hpp, template parameter E).
Same structure, three more mantissa bits: every pick must be near-best within the float16 rounding of the operands,
the float32 BMUs must be missed less often than in bf16, the update path is the shared exact-f32 one, and rows or
units outside the float16 range are refused.  GPU only (`-m gpu`)."""
import numpy as np
import pytest

from oracle import som_oracle as O

pytestmark = pytest.mark.gpu
F32 = np.float32


def engine(X, Y, D, **kw):
    from xpysom_dask_amd.engine import HipEngine
    return HipEngine(X, Y, D, **kw)


def rel_err(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


# resident (D <= 128), wide (K >= 4096, D <= 800), tiled (small map / D > 800)
SHAPES = [(20, 24, 128, 5000, "f16"), (30, 30, 17, 3001, "f16"), (64, 66, 200, 2500, "f16"), (70, 64, 784, 1500, "f16"),
          (9, 9, 300, 700, "f16"), (64, 64, 900, 600, "f16")]


@pytest.mark.parametrize("X,Y,D,n,prec", SHAPES)
def test_f16_picks_are_near_best_and_closer_to_float32_than_bf16(X, Y, D, n, prec):
    data = O.gaussian_blobs(n, D, seed=D + n)
    w = (np.random.RandomState(X * Y).rand(X, Y, D) * 2 - 1).astype(F32) * 2
    wf = w.reshape(-1, D)
    ref = O.bmu_ids(data, wf)
    got = {}
    for p in (prec, prec.replace("f16", "bf16")):
        e = engine(X, Y, D, precision=p)
        e.set_weights(w)
        e.set_data(data)
        e.epoch_accumulate(3.0, 0.4, True)
        num, den, bmu = e.epoch_fetch()
        assert np.array_equal(e.bmu(data[:200]), bmu[:200])          # query grid == resident grid
        got[p] = bmu
        if p == prec:
            _, onum, oden = O.update(data, w, 0.4, 3.0, wide=True, forced_bmu=bmu)
            assert rel_err(num, onum.reshape(-1, D)) < 1e-5 and rel_err(den, oden.reshape(-1)) < 1e-5
        e.close()
    bmu = got[prec]
    x64, w64 = data.astype(np.float64), wf.astype(np.float64)
    dd = np.sqrt(np.maximum((x64 ** 2).sum(1)[:, None] - 2 * x64 @ w64.T + (w64 ** 2).sum(1)[None, :], 0))
    slack = (2.0 ** -11 if prec == "f16" else 2.0 ** -15) * (np.linalg.norm(x64, axis=1) + np.linalg.norm(w64, axis=1).max())
    assert (dd[np.arange(n), bmu] <= dd.min(1) + slack).all()
    miss_f16, miss_bf16 = (bmu != ref).sum(), (got[prec.replace("f16", "bf16")] != ref).sum()
    assert miss_f16 <= miss_bf16
    if prec == "f16":
        assert miss_f16 <= max(2, n // 300)                            # bf16 misses ~0.5 % of these rows


@pytest.mark.parametrize("shape,dist", [((20, 24, 96), "euclidean"), ((64, 64, 200), "cosine"), ((64, 66, 133), "euclidean")])
def test_f16_fused_merge_equals_separate_launches_on_the_first_epoch(shape, dist, monkeypatch):
    X, Y, D = shape
    n = 3000
    data = np.abs(O.gaussian_blobs(n, D, seed=11))
    w = np.abs(O.default_codebook(X, Y, D, 9).astype(F32))
    outs = []
    for fuse in ("0", "1"):
        monkeypatch.setenv("SOM_FUSE_MERGE", fuse)
        e = engine(X, Y, D, precision="f16", distance=dist)
        e.set_weights(w)
        e.set_data(data)
        trace = []
        for t, (sig, eta) in enumerate([(6.0, 0.5), (1.5, 0.2), (0.4, 0.05)]):
            e.epoch_accumulate(sig, eta, True)
            num, den, bmu = e.epoch_fetch()
            e.epoch_merge()
            trace.append((bmu, e.get_weights(), num, den))
        outs.append(trace)
        e.close()
    (b0, w0, n0, d0), (b1, w1, n1, d1) = outs[0][0], outs[1][0]
    assert np.array_equal(b0, b1) and np.array_equal(n0, n1) and np.array_equal(d0, d1) and np.array_equal(w0, w1)
    for (b0, w0, _, _), (b1, w1, _, _) in zip(outs[0][1:], outs[1][1:]):
        assert (b0 != b1).mean() < 0.01


def test_f16_refuses_rows_and_units_outside_the_float16_range():
    from xpysom_dask_amd.engine import SomHipError
    X, Y, D = 6, 6, 8
    e = engine(X, Y, D, precision="f16")
    e.set_weights(np.ones((X, Y, D), F32))
    big = np.full((10, D), 7.0e4, F32)
    with pytest.raises(SomHipError, match="float16"):
        e.set_data(big)
    with pytest.raises(SomHipError, match="float16"):
        e.set_weights(np.full((X, Y, D), 7.0e4, F32))
    e.set_data(np.full((10, D), 100.0, F32))                           # inside the range: accepted
    ec = engine(X, Y, D, precision="f16", distance="cosine")           # cosine rounds unit-length rows: any magnitude
    ec.set_weights(np.full((X, Y, D), 7.0e4, F32))
    ec.set_data(np.abs(big))
    assert ec.bmu(np.abs(big)).shape == (10,)


def test_f16_saturates_instead_of_overflowing():
    """Query rows are not range-checked (no host round trip on that path): a component beyond 65504 saturates in the
    float16 image instead of becoming infinite -- an infinite row norm would make the launch's offset B, and with it
    every other row's distances, infinite."""
    X, Y, D, n = 12, 12, 16, 600
    data = O.gaussian_blobs(n, D, seed=4)
    w = (np.random.RandomState(1).rand(X, Y, D) * 2 - 1).astype(F32)
    ref = O.bmu_ids(data, w.reshape(-1, D))
    e = engine(X, Y, D, precision="f16")
    e.set_weights(w)
    q = data.copy()
    q[5, 3] = 1.0e6                                                    # float16(1e6) = inf without the saturation
    got = e.bmu(q)
    assert ((got >= 0) & (got < X * Y)).all()
    keep = np.arange(n) != 5
    # (the saturated row raises B to ~65504 |w|max, so the others are compared at a coarser absolute resolution)
    assert (got[keep] == ref[keep]).mean() > 0.5
    clamped = q.copy()
    clamped[5, 3] = 65504.0
    assert got[5] == O.bmu_ids(clamped[5:6], w.reshape(-1, D))[0]


def test_f16_class_surface_trains_and_scores_like_float32():
    from xpysom_dask_amd import XPySom
    data = O.gaussian_blobs(20000, 24, seed=2)
    qe = {}
    for prec in ("f32", "f16"):
        som = XPySom(24, 20, 24, sigma=4.0, learning_rate=0.5, random_seed=7, precision=prec)
        som.train(data, 6)
        qe[prec] = som.quantization_error(data[:5000])
        assert len(som.winner(data[:9])) == 9
    assert abs(qe["f16"] - qe["f32"]) < 2e-3 * qe["f32"]
