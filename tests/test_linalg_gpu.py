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
