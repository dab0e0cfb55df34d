"""Sparse (dgCMatrix-style CSC) input: the block is expanded on the device to the same dense fp32 layout, so every result must be
bit-identical to the dense entry points (reference: R/SHARP.R:343-345,579 accept whatever log2(scExp + 1) and %*% accept)."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sa():
    import sharp_amd

    sharp_amd.init(0)
    return sharp_amd


def _counts(oracle, n, m, G=5, seed=77):
    return oracle.synth_fill(seed, m, 0, n, G, max(50, m // (2 * G)))


@pytest.mark.parametrize("n", [700, 6100])
def test_sparse_input_equals_dense_input(sa, oracle, n):
    X = _counts(oracle, n, 1800)
    dense = sa.SHARP(X, ensize_K=3, rN_seed=11, logflag=False)
    for fmt in (sp.csc_matrix, sp.csr_matrix, sp.coo_matrix):
        res = sa.SHARP(fmt(X), ensize_K=3, rN_seed=11, logflag=False)
        assert np.array_equal(res["pred_clusters"], dense["pred_clusters"])
        assert np.array_equal(res["viE"], dense["viE"])                     # same dense block in HBM -> same bits
        assert np.array_equal(res["x0"], dense["x0"])
    ref = oracle.SHARP(X, K=3, rN_seed=11, nthreads=8)
    assert np.array_equal(dense["pred_clusters"], ref["pred_clusters"])


def test_sparse_front_door_prep_cpm_and_testlog(sa, oracle):
    X = _counts(oracle, 600, 1500)
    X[5] = 0.0                         # an all-zero gene: removed by prep (R/SHARP.R:104-106)
    X[7, 3] = -2.0                     # a negative value: replaced by 0 with a warning (:100-103)
    cells = np.arange(0, 600, 6)
    with pytest.warns(UserWarning):
        dense = sa.SHARP(X, exp_type="count", ensize_K=3, rN_seed=5, testlog_cells=cells)
    with pytest.warns(UserWarning):
        res = sa.SHARP(sp.csc_matrix(X), exp_type="count", ensize_K=3, rN_seed=5, testlog_cells=cells)
    assert res["N.genes"] == dense["N.genes"] == 1500
    assert res["paras"]["logmark"] == dense["paras"]["logmark"]
    assert np.array_equal(res["pred_clusters"], dense["pred_clusters"])
    np.testing.assert_allclose(res["viE"], dense["viE"], rtol=0, atol=1e-9 * np.abs(dense["viE"]).max())


def test_csc_to_dev_and_duplicates_and_empty_columns(sa, oracle):
    import torch
    from sharp_amd import device as dev

    X = _counts(oracle, 300, 1001)     # odd gene count: column stride != multiple of 4 on the caller's side
    X[:, 17] = 0.0                     # an empty cell column
    dX = dev.csc_to_dev(sp.csc_matrix(X))
    assert dX.shape == (300, 1001)
    assert np.array_equal(dX.cpu().numpy(), X.T.astype(np.float32))
    # duplicated entries are summed first (scipy semantics; a dgCMatrix never holds any)
    r = np.array([0, 0, 3]); c = np.array([1, 1, 2]); v = np.array([2.0, 3.0, 4.0])
    d2 = dev.csc_to_dev(sp.coo_matrix((v, (r, c)), shape=(5, 4)))
    exp = np.zeros((4, 5), np.float32); exp[1, 0] = 5; exp[2, 3] = 4
    assert np.array_equal(d2.cpu().numpy(), exp)
    del torch


def test_sparse_input_errors(sa):
    from sharp_amd import _lib

    lib = _lib.lib()
    import torch

    dX = torch.zeros((3, 8), dtype=torch.float32, device="cuda")
    cp = np.array([0, 1, 2, 3], np.int32); ri = np.array([0, 9, 1], np.int32); xv = np.ones(3)
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    rc = lib.sharp_csc_to_dense_dev(ip(cp), ip(ri), dp(xv), 8, C.c_longlong(3), C.c_void_p(dX.data_ptr()), C.c_longlong(8))
    assert rc != 0 and b"row index" in lib.sharp_last_error()
    assert float(dX.abs().sum()) == 2.0                       # the two valid entries were written, the bad one never
    cp2 = np.array([0, 2, 1, 3], np.int32)
    rc = lib.sharp_csc_to_dense_dev(ip(cp2), ip(ri), dp(xv), 8, C.c_longlong(3), C.c_void_p(dX.data_ptr()), C.c_longlong(8))
    assert rc != 0 and b"non-decreasing" in lib.sharp_last_error()
