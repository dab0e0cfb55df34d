"""The fp64 MFMA GEMM kernels (sharp_amd/csrc/linalg.hip) against numpy, directly: ragged M / N / K (K not a multiple of the 16-deep
k tile, M and N not multiples of the 64- and 128-wide tiles), both kernels, the three epilogues, the symmetric (mirrored) form."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    import sharp_amd

    sharp_amd.init(0)
    return sharp_amd.lib()


def _gemm(lib, At, Bt, epilogue=0, symmetric=0, fast=0):
    K, M = At.shape
    N = M if symmetric else Bt.shape[1]
    Cm = np.full((M, N), np.nan)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    rc = lib.sharp_gemm_tn_f64(dp(At), dp(At if symmetric else Bt), dp(Cm), M, N, K, epilogue, symmetric, fast)
    assert rc == 0, lib.sharp_last_error()
    return Cm


@pytest.mark.parametrize("fast", [0, 1])
@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (63, 65, 17), (64, 64, 16), (130, 127, 391), (257, 40, 5), (300, 513, 33), (128, 128, 474)])
def test_plain_product(lib, fast, M, N, K):
    rng = np.random.default_rng(M * 1000 + N + K)
    At, Bt = rng.standard_normal((K, M)), rng.standard_normal((K, N))
    got = _gemm(lib, At, Bt, fast=fast)
    ref = At.T @ Bt
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-13 * K * max(1.0, np.abs(ref).max()))


@pytest.mark.parametrize("fast", [0, 1])
@pytest.mark.parametrize("n,K", [(5, 3), (129, 100), (200, 391), (385, 16)])
def test_symmetric_correlation_epilogues(lib, fast, n, K):
    rng = np.random.default_rng(n + K)
    U = rng.standard_normal((K, n))
    U -= U.mean(0)
    U /= np.linalg.norm(U, axis=0)                      # unit centred columns: U^T U is a correlation matrix
    cor = np.clip(U.T @ U, -1, 1)
    d = _gemm(lib, U, U, epilogue=1, symmetric=1, fast=fast)
    ref = 1 - cor
    np.fill_diagonal(ref, 0.0)
    np.testing.assert_allclose(d, ref, rtol=0, atol=1e-13 * K)
    assert np.array_equal(d, d.T) and np.all(np.diag(d) == 0.0)        # mirrored, exact diagonal
    sres = _gemm(lib, U, U, epilogue=2, symmetric=1, fast=fast)
    ref2 = cor.copy()
    np.fill_diagonal(ref2, 1.0)
    np.testing.assert_allclose(sres, ref2, rtol=0, atol=1e-13 * K)
    assert np.array_equal(sres, sres.T) and np.all(np.diag(sres) == 1.0) and sres.max() <= 1.0


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
