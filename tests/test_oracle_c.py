"""The scalar C restatement (oracle/som_oracle.c) against NumPy and the golden vectors.
It states the float32 operation ORDER explicitly (k-sequential fma chain, NumPy pairwise
sums); these tests show that order reproduces the reference's distance matrix bit for bit,
which is what lets the f32 MFMA kernel match BMUs exactly, near-ties included.  CPU only."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import som_oracle as O
from tests.conftest import REPO, load_golden

F32 = np.float32
SO = os.path.join(REPO, "oracle", "_build", "libsomoracle.so")


@pytest.fixture(scope="module")
def lib():
    subprocess.check_call(["make", "-s", "-C", os.path.join(REPO, "oracle")])
    return C.CDLL(SO)


def fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def c_row_sq(lib, a):
    a = np.ascontiguousarray(a, F32)
    out = np.empty(len(a), F32)
    lib.oracle_row_sq_f32(fp(a), C.c_long(len(a)), C.c_int(a.shape[1]), fp(out))
    return out


def c_bmu(lib, x, w):
    x, w = np.ascontiguousarray(x, F32), np.ascontiguousarray(w, F32)
    wsq = c_row_sq(lib, w)
    ids = np.empty(len(x), np.int32)
    lib.oracle_bmu_euclid_f32(fp(x), fp(w), fp(wsq), C.c_long(len(x)), C.c_long(len(w)), C.c_int(x.shape[1]), ip(ids))
    return ids


@pytest.mark.parametrize("D", [1, 3, 7, 8, 9, 31, 32, 100, 128, 129, 300, 784])
def test_row_sq_is_numpy_bitwise(lib, D):
    a = np.random.RandomState(D).randn(40, D).astype(F32)
    assert np.array_equal(c_row_sq(lib, a), np.power(a, 2).sum(axis=1))


@pytest.mark.parametrize("shape", [(150, 36, 4), (512, 576, 16), (256, 4096, 32), (64, 4096, 128)])
def test_fma_chain_is_the_sgemm_of_this_numpy_build(lib, shape):
    """Documents the observation the bit-exact BMU claim rests on (OpenBLAS 0.3.29, this image):
    for one K block (D <= 448) sgemm == k-sequential fmaf chain from 0."""
    N, K, D = shape
    rs = np.random.RandomState(1)
    x, w = rs.randn(N, D).astype(F32), rs.randn(K, D).astype(F32)
    out = np.empty((N, K), F32)
    lib.oracle_cross_f32(fp(x), fp(w), C.c_long(N), C.c_long(K), C.c_int(D), fp(out))
    assert np.array_equal(out, np.dot(x, w.T))


@pytest.mark.parametrize("shape", ["6x6x4", "8x8x3", "24x24x16", "20x30x12"])
@pytest.mark.parametrize("decay", ["linear", "exponential"])
def test_c_hot_loop_against_golden(lib, shape, decay):
    g = load_golden("g4_update_" + shape)
    X, Y, D, n = (int(v) for v in g["shape"])
    data = O.gaussian_blobs(n, D, seed=int(g["data_seed"]))
    wide = int(O.decay_is_wide(decay))
    w0 = O.default_codebook(X, Y, D, 1234).astype(F32)
    for tag, w in (("init", w0), ("mid", g[f"{decay}_wmid"]), ("last", g[f"{decay}_wmid"])):
        wf = np.ascontiguousarray(w.reshape(-1, D))
        bmu = c_bmu(lib, data, wf)
        assert np.array_equal(bmu, g[f"{decay}_{tag}_bmu"])            # bit-exact, near-ties included
        num = np.empty((X * Y, D), np.float64)
        den = np.empty(X * Y, np.float64)
        lib.oracle_update_gaussian(fp(data), ip(bmu), C.c_long(n), C.c_int(X), C.c_int(Y), C.c_int(D),
                                   C.c_double(float(g[f"{decay}_{tag}_sig"])), C.c_double(float(g[f"{decay}_{tag}_eta"])),
                                   C.c_double(0.5), C.c_int(wide), dp(num), dp(den))
        gden = g[f"{decay}_{tag}_den"].reshape(-1)
        ok = gden > 1e-30
        np.testing.assert_allclose(den[ok], gden[ok], rtol=1e-5)
        if f"{decay}_{tag}_num" in g:
            np.testing.assert_allclose(num, g[f"{decay}_{tag}_num"].reshape(-1, D), rtol=1e-5,
                                       atol=1e-6 * np.abs(num).max())
        wnew = np.array(wf)
        lib.oracle_merge_f32(fp(wnew), fp(np.ascontiguousarray(num, F32)), fp(np.ascontiguousarray(den, F32)),
                             C.c_long(X * Y), C.c_int(D))
        gw = g[f"{decay}_{tag}_wout"].reshape(-1, D)
        np.testing.assert_allclose(wnew[ok], gw[ok], rtol=1e-5, atol=1e-5 * np.abs(gw).max())


def test_c_bmu_on_exact_ties(lib):
    g = load_golden("g1_ties")
    x, w = g["x"].astype(F32), g["w"].astype(F32)
    assert np.array_equal(c_bmu(lib, x, w.reshape(-1, w.shape[2])), g["ids"])
    assert np.array_equal(c_bmu(lib, x, np.zeros((20, 8), F32)), g["ids_zero"])
