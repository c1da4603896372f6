"""Host-side behaviour that needs no GPU: constructor validation and error strings
(xpysom.py:164-165,196-198,217-220,228-231; distances.py:172-175), the seeded default
codebook, schedules, the C-ABI library's exported surface, fail-loud without a device."""
import ctypes as C
import pickle
import re
import warnings

import numpy as np
import pytest

from oracle import som_oracle as O
from tests.conftest import REPO, _gpu_present


def test_constructor_mirrors_reference_validation():
    from xpysom_dask_amd import XPySom
    with pytest.raises(ValueError, match="boooom not supported. Functions available"):
        XPySom(5, 5, 1, neighborhood_function='boooom')
    with pytest.raises(ValueError, match="ridethewave not supported. Distances available"):
        XPySom(5, 5, 1, activation_distance='ridethewave')
    with pytest.raises(ValueError, match="not supported only hexagonal and rectangular available"):
        XPySom(5, 5, 1, topology='triangular')
    with pytest.raises(ValueError, match="sqrt not supported. Functions available"):
        XPySom(5, 5, 1, decay_function='sqrt')
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        XPySom(5, 5, 1, sigma=5)
        assert any("sigma is too high" in str(x.message) for x in w)
    som = XPySom(6, 4, 3)
    assert som._sigma == 2.0 and som._n_parallel == 65536
    with pytest.raises(ValueError, match=r"Received 2 features, expected 3\."):
        som.quantization_error([[1, 2]])
    for name in ('manhattan', 'manhattan_no_opt', 'norm_p', 'norm_p_no_opt', 'cosine', 'euclidean_no_opt'):
        XPySom(5, 5, 1, activation_distance=name)                      # the whole distances.py registry
    with pytest.raises(ValueError, match="triangle not supported"):
        XPySom(5, 5, 1, topology='hexagonal', neighborhood_function='triangle')


def test_default_codebook_is_the_reference_formula():
    from xpysom_dask_amd import XPySom
    som = XPySom(7, 5, 3, random_seed=1234)
    np.testing.assert_array_equal(som._weights, O.default_codebook(7, 5, 3, 1234))
    assert som._weights.dtype == np.float64
    np.testing.assert_allclose(np.linalg.norm(som._weights, axis=-1), 1.0, atol=1e-12)


def test_schedules_equal_the_oracle_bit_for_bit():
    from xpysom_dask_amd.decays import DECAY_FUNCTIONS
    for name, f in DECAY_FUNCTIONS.items():
        g = O.DECAYS[name]
        for T in (1, 10, 100):
            for t in range(0, T, max(1, T // 7)):
                for v0, vN in ((8.0, 1), (0.5, 0.01), (3.0, 0)):
                    a, b = f(v0, vN, t, T), g(v0, vN, t, T)
                    assert a == b and type(a) is type(b)


def test_synthetic_generator_is_the_oracles():
    from xpysom_dask_amd.synthetic import gaussian_blobs
    np.testing.assert_array_equal(gaussian_blobs(100, 7, seed=5), O.gaussian_blobs(100, 7, seed=5))


def test_c_abi_exports_every_declared_symbol():
    from xpysom_dask_amd import _lib
    lib = _lib.load()
    # (the boundary, include/somhip.h, + the test-only entry points of include/somhip_test.h)
    header = open(REPO + "/include/somhip.h").read() + open(REPO + "/include/somhip_test.h").read()
    declared = set(re.findall(r"\b(som_[a-z0-9_]+)\s*\(", header))
    assert not re.search(r"\bsom_debug_[a-z0-9_]+\s*\(", open(REPO + "/include/somhip.h").read()), "test hooks belong in somhip_test.h"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name)
    assert lib.som_version().startswith(b"somhip")


@pytest.mark.parametrize("X,Y", [(256, 256), (16, 24), (100, 100), (13, 21), (9, 100), (70, 3), (1, 1), (1, 200), (5, 5)])
def test_exact_patch_order_is_a_compact_ascending_bijection(X, Y):
    """som_patch_order (host arithmetic of the exact mode): a bijection position -> unit; a group of 64 positions
    is an 8 x 8 patch of the map where both sides are multiples of 8 -- held as four 4 x 4 blocks, the plan's 16-unit sub-blocks
    (the re-score decides equal scores by rank there) -- and a run of whole 8-row bands, unit ids ascending, anywhere else."""
    from xpysom_dask_amd import _lib
    lib = _lib.load()
    K = X * Y
    perm = np.full(K, -1, np.int32)
    assert lib.som_patch_order(X, Y, perm.ctypes.data_as(C.POINTER(C.c_int32))) == 0
    assert np.array_equal(np.sort(perm), np.arange(K))
    for g in range(0, K, 64):
        grp = perm[g:g + 64]
        xs, ys = grp // Y, grp % Y
        if X % 8 == 0 and Y % 8 == 0:
            for b in range(4):                                                  # block b: rows 4 (b >> 1).., columns 4 (b & 1)..
                bx, by = xs[16 * b:16 * b + 16] - xs.min(), ys[16 * b:16 * b + 16] - ys.min()
                assert np.array_equal(bx, 4 * (b >> 1) + np.arange(16) // 4) and np.array_equal(by, 4 * (b & 1) + np.arange(16) % 4)
        else:
            assert (np.diff(grp) > 0).all()
        assert xs.max() - xs.min() < 8 * (-(-64 // (8 * Y)) + 1)            # whole bands of 8 map rows, one more when it straddles
        if X % 8 == 0 and Y % 8 == 0:
            assert xs.max() - xs.min() == 7 and ys.max() - ys.min() == 7 and xs.min() % 8 == 0 and ys.min() % 8 == 0
    assert lib.som_patch_order(0, 4, perm.ctypes.data_as(C.POINTER(C.c_int32))) != 0


@pytest.mark.skipif(_gpu_present(), reason="checks the no-GPU failure mode")
def test_compute_fails_loudly_without_a_gpu():
    from xpysom_dask_amd import XPySom
    from xpysom_dask_amd.engine import SomHipError
    som = XPySom(4, 4, 2, random_seed=0)
    with pytest.raises(SomHipError, match="no HIP device"):
        som.train(np.zeros((8, 2)), 1)
    with pytest.raises(SomHipError):
        som.winner(np.zeros((3, 2)))


@pytest.fixture
def oracle_engine(monkeypatch):
    """The engine class the host code constructs, swapped for the oracle-backed test double."""
    from tests.oracle_engine import OracleEngine
    from xpysom_dask_amd import engine
    monkeypatch.setattr(engine, "HipEngine", OracleEngine)


def test_host_logic_end_to_end_with_the_test_double(oracle_engine):
    """XPySom's epoch loop / schedule plumbing / result formatting, engine replaced by the oracle."""
    from xpysom_dask_amd import XPySom
    data = O.gaussian_blobs(300, 4, seed=2)
    for decay in ("linear", "exponential", "asymptotic"):
        som = XPySom(6, 5, 4, random_seed=9, decay_function=decay)
        w0 = som._weights.copy()
        assert som.train(data, 7) is som
        ref = O.train(data, w0, 7, sigma0=2.5, decay=decay)
        np.testing.assert_allclose(som._weights, ref, rtol=1e-6, atol=1e-7)
        # resume: 0..3 then 3..7 equals one run
        som2 = XPySom(6, 5, 4, random_seed=9, decay_function=decay)
        som2.train(data, 7, iter_beg=0, iter_end=3)
        som2.train(data, 7, iter_beg=3)
        np.testing.assert_array_equal(som2._weights, som._weights)
    w = som.winner(data[:5])
    assert isinstance(w, list) and len(w) == 5 and isinstance(w[0][0], np.int64)
    assert isinstance(som.winner(data[0]), tuple)
    blob = pickle.dumps(som)
    back = pickle.loads(blob)
    np.testing.assert_array_equal(back._weights, som._weights)
    assert back._engine_obj is None


def test_train_streaming_host_logic_with_the_test_double(oracle_engine):
    from xpysom_dask_amd import XPySom
    data = O.gaussian_blobs(500, 4, seed=12)
    a = XPySom(6, 5, 4, random_seed=9, decay_function="linear")
    b = XPySom(6, 5, 4, random_seed=9, decay_function="linear")
    a.train(data, 4)
    b.train_streaming(lambda: (data[i:i + 123] for i in range(0, 500, 123)), 4)
    np.testing.assert_array_equal(a._weights, b._weights)


def test_device_rows_detection_and_validation():
    """Host arrays are not device rows; a __cuda_array_interface__ object must be C-contiguous float32 2-D."""
    from xpysom_dask_amd.xpysom import _device_rows

    class Fake:
        def __init__(self, shape, typestr="<f4", strides=None):
            self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (4096, False),
                                             "version": 2, "strides": strides}
    assert _device_rows(np.zeros((3, 2), dtype=np.float32)) is None
    assert _device_rows([[1.0, 2.0]]) is None
    ptr, n, d, dev, owner, stream = _device_rows(Fake((7, 5)))
    assert (ptr, n, d, dev, stream) == (4096, 7, 5, None, None) and isinstance(owner, Fake)
    f = Fake((7, 5))
    f.__cuda_array_interface__["stream"] = 1                  # the producer's stream travels with the rows
    assert _device_rows(f)[5] == 1
    for bad in (Fake((7,)), Fake((7, 5), typestr="<f8"), Fake((7, 5), strides=(40, 4))):
        with pytest.raises(ValueError):
            _device_rows(bad)
    import torch
    assert _device_rows(torch.zeros(4, 3)) is None          # CPU tensor: host path


def test_distance_map_matches_the_reference():
    """The vectorised U-matrix against the reference's own output (xpysom.py:788-817), both topologies,
    non-square and degenerate (1 x Y) maps -- host-only, no engine involved."""
    from tests.conftest import load_golden
    from xpysom_dask_amd import XPySom
    g = load_golden("g14_distance_map")
    for topo in ("rectangular", "hexagonal"):
        for (X, Y, D) in ((7, 6, 3), (4, 9, 5), (1, 5, 2)):
            som = XPySom(X, Y, D, random_seed=31, topology=topo)
            np.testing.assert_allclose(som.distance_map(), g[f"{topo}_{X}x{Y}x{D}"], rtol=1e-12, atol=0)


def test_constructor_refuses_what_the_engine_would_refuse():
    from xpysom_dask_amd import XPySom
    with pytest.raises(ValueError, match="mexican_hat with compact_support needs a square map"):
        XPySom(5, 7, 3, neighborhood_function="mexican_hat", compact_support=True)
    XPySom(5, 5, 3, neighborhood_function="mexican_hat", compact_support=True)
    XPySom(5, 7, 3, neighborhood_function="mexican_hat", compact_support=True, topology="hexagonal")
    for dist in ("manhattan", "norm_p", "norm_p_no_opt", "euclidean_no_opt"):
        with pytest.raises(ValueError, match="needs precision='f32'"):
            XPySom(5, 5, 3, activation_distance=dist, precision="bf16")
    for p in (0, 17, "2", -1.5, 0.0, float("nan"), float("inf"), 64.5):
        with pytest.raises(NotImplementedError, match="norm_p"):
            XPySom(5, 5, 3, activation_distance="norm_p", activation_distance_kwargs={"p": p})
    for p in (3.0, 1.5, 0.5, np.float32(2.5), 63.5):     # integers as floats, and real exponents (distances.py:61-75)
        XPySom(5, 5, 3, activation_distance="norm_p", activation_distance_kwargs={"p": p})
    XPySom(5, 5, 3, activation_distance="norm_p_no_opt")


def test_win_map_labels_map_activation_response_equal_the_per_sample_definition(oracle_engine):
    """The batched grouping (one BMU call + a stable sort) against the reference's per-sample definition
    (xpysom.py:819-865: winner(x) for every x, appended in data order)."""
    from collections import Counter, defaultdict
    from xpysom_dask_amd import XPySom
    data = O.gaussian_blobs(400, 4, seed=8)
    labels = [int(v) % 3 for v in np.arange(400) * 7]
    som = XPySom(5, 6, 4, random_seed=2)
    wins = som.winner(data)
    want_w, want_l = defaultdict(list), defaultdict(list)
    for x, w_, lab in zip(data, wins, labels):
        want_w[w_].append(x)
        want_l[w_].append(lab)
    wm, lm = som.win_map(data), som.labels_map(data, labels)
    assert list(wm) == list(want_w) == list(lm)                   # units in order of first win
    for k in want_w:
        assert all(np.array_equal(a, b) for a, b in zip(wm[k], want_w[k])) and len(wm[k]) == len(want_w[k])
        assert lm[k] == Counter(want_l[k])
    resp = som.activation_response(data)
    assert resp.shape == (5, 6) and resp.sum() == 400 and all(resp[k] == len(v) for k, v in want_w.items())
    with pytest.raises(ValueError, match="same length"):
        som.labels_map(data, labels[:-1])


def test_mexican_hat_with_a_python_float_sigma_of_zero_raises_as_the_reference_does():
    """neighborhoods.py:72 / :94: `1 - 2/d*p` with d = 2 std_coeff^2 sigma^2 -- a linear schedule that ends at sigmaN=0
    hands the last epoch a Python float 0.0 and the reference raises ZeroDivisionError there (oracle/diff_host_reference.py
    found it); a numpy.float64 zero (exponential decay cannot produce one) divides to inf instead."""
    import numpy as np
    from xpysom_dask_amd import XPySom
    som = XPySom(4, 4, 2, neighborhood_function="mexican_hat", decay_function="linear", sigma=1.0, sigmaN=0)
    with pytest.raises(ZeroDivisionError, match="float division by zero"):
        som._check_sigma(0.0)
    som._check_sigma(np.float64(0.0))
    som._check_sigma(0.5)
    XPySom(4, 4, 2, neighborhood_function="gaussian")._check_sigma(0.0)      # gaussian: 0/0 -> NaN, no exception



def test_comm_load_of_a_missing_library_fails_with_a_message():
    """som_comm_load(path) means that library and no other: a path that does not exist returns non-zero and leaves a
    'librccl not found' message (round 2 read dlerror() twice there: a NULL string concatenation)."""
    from xpysom_dask_amd import _lib
    lib = _lib.load()
    rc = lib.som_comm_load(b"/nonexistent/dir/librccl-not-here.so")
    msg = lib.som_last_error(None).decode()
    if rc == 0:
        pytest.skip("a RCCL library was already bound in this process")
    assert rc != 0 and "librccl not found" in msg and "nonexistent" in msg


def test_train_keeps_the_completed_epochs_when_an_epoch_raises(oracle_engine):
    """mexican_hat + a linear schedule ending at sigmaN=0: the last epoch raises ZeroDivisionError (as the reference's
    does); the object must then hold the codebook of the epochs that completed, not its pre-train codebook."""
    from xpysom_dask_amd import XPySom
    data = O.gaussian_blobs(200, 3, seed=1)
    som = XPySom(5, 5, 3, neighborhood_function="mexican_hat", decay_function="linear", sigma=2.0, sigmaN=0, random_seed=1)
    w0 = som._weights.copy()
    with pytest.raises(ZeroDivisionError):
        som.train(data, 4)
    ref = XPySom(5, 5, 3, neighborhood_function="mexican_hat", decay_function="linear", sigma=2.0, sigmaN=0, random_seed=1)
    ref.train(data, 4, iter_beg=0, iter_end=3)          # the three epochs that run before sigma reaches 0
    assert not np.array_equal(som._weights, w0.astype(np.float32))
    np.testing.assert_array_equal(som._weights, ref._weights)


def test_precision_exact_is_accepted_with_every_distance():
    from xpysom_dask_amd import XPySom
    for dist in ("euclidean", "cosine", "manhattan", "norm_p"):
        XPySom(4, 4, 3, activation_distance=dist, precision="exact")
    with pytest.raises(ValueError):
        XPySom(4, 4, 3, activation_distance="manhattan", precision="bf16")
