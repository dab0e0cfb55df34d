"""LAB build only: the correlation-distance GEMM on the integer matrix cores (tools/lab/gemm_i8.hip, SHARP_DIST_I8=1; tools/lab/sharp_lab.h)."""
import ctypes as C

import numpy as np
import pytest

SEED = 20261003


@pytest.fixture(scope="module")
def sa():
    import sharp_amd

    sharp_amd.init(0)
    return sharp_amd


@pytest.fixture(scope="module")
def lib(sa):
    return sa.lib()


@pytest.mark.parametrize("n,p", [(3, 2), (64, 8), (129, 33), (300, 50), (1000, 223), (2000, 391), (1999, 474)])
def test_distance_matrix_on_the_integer_matrix_cores(lib, n, p):
    """gemm_i8.hip (SHARP_DIST_I8=1): D = 1 - U U^T of centred unit rows through seven 7-bit digits per entry and exact int8 products.
    Against a long-double product: the digits truncate a row below 2^-49 of its largest entry, so |D - ref| stays below 5e-14 (seven
    digits: 1-2e-14 measured; the fp64 MFMA kernel has 2e-15); symmetric, zero diagonal, ragged n and p (n not a multiple of the 32-row
    blocks, p not a multiple of the 32-deep k step), rows of very different scale (their powers of two differ)."""
    rng = np.random.default_rng(n * 7 + p)
    X = rng.standard_normal((n, p)) * np.exp(2 * rng.standard_normal((n, 1)))
    X[: n // 3] += 3 * rng.standard_normal((1, p))
    X[n // 2, :] = 0.0
    X[n // 2, p // 2] = 5.0                                   # one entry dominates its row
    Xc = X - X.mean(1, keepdims=True)
    U = np.ascontiguousarray(Xc / np.sqrt((Xc * Xc).sum(1, keepdims=True)))
    D = np.full((n, n), np.nan)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    rc = lib.sharp_dist_i8(dp(U), n, p, dp(D))
    assert rc == 0, lib.sharp_last_error()
    Ul = U.astype(np.longdouble)
    ref = 1 - np.clip(Ul @ Ul.T, -1, 1)
    np.fill_diagonal(ref, 0)
    assert np.abs(D - ref.astype(np.float64)).max() < 5e-14
    assert np.array_equal(D, D.T) and not D.diagonal().any()


def test_labels_with_the_integer_distance_gemm(sa, oracle, monkeypatch):
    """SHARP_DIST_I8=1: the distance matrices of the base clustering through gemm_i8.hip (exact int8 products of 7-bit digits) instead
    of the fp64 MFMA kernel: same labels as the oracle, SHARP_small (one task of all cells) and SHARP_large (ragged folds)."""
    monkeypatch.setenv("SHARP_DIST_I8", "1")
    for n, m, K, seed in ((1500, 1800, 5, 11), (7300, 2100, 3, 12)):
        X = oracle.synth_fill(20261004 + seed, m, 0, n, 5, 120)
        ref = oracle.SHARP(X, K=K, rN_seed=77, nthreads=8)
        res = sa.SHARP(X, ensize_K=K, rN_seed=77, forview=False, logflag=False)
        assert np.array_equal(res["pred_clusters"], ref["pred_clusters"]), (n, m, K)
