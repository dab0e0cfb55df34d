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


def test_wire_formats_of_host_blocks(sa, oracle):
    """What crosses PCIe (sharp_x_wire): counts as unsigned 8- or 16-bit integers, other fp32-exact values as floats, anything else as doubles
    -- decided while the block is packed -- and 16-bit row indices for a sparse block of at most 65 536 genes.  Whatever the wire, the block
    in HBM is the same, so every result equals the dense fp64 reference path: labels and projections against the oracle."""
    lib = sa.lib()
    X = _counts(oracle, 640, 1500)
    kinds = []
    for scale, wire, storage in ((1.0, 16, 32), (-1.0, 8, 32), (0.5, 32, 32), (1.0 / 3.0, 64, 64)):
        for fmt in (np.asfortranarray, sp.csc_matrix):
            Xs = X * abs(scale)
            if scale == 1.0:
                Xs = Xs.copy(); Xs[3, 5] = 65535.0                       # the largest value the 16-bit wire holds
            if scale == -1.0:
                Xs = np.minimum(Xs, 255.0)                               # counts up to 255: 8-bit values
            res = sa.SHARP(fmt(Xs), ensize_K=3, rN_seed=11, logflag=False, prep=False)
            assert (lib.sharp_x_wire(), lib.sharp_x_storage()) == (wire, storage), (scale, fmt)
            r2 = oracle.SHARP(Xs, K=3, rN_seed=11, nthreads=4, want_view=True)
            assert np.array_equal(res["pred_clusters"], r2["pred_clusters"])
            np.testing.assert_allclose(res["viE"], r2["viE"], rtol=0, atol=2e-12 * np.abs(r2["viE"]).max())
            kinds.append(res["viE"])
    for q in range(0, len(kinds), 2):
        np.testing.assert_array_equal(kinds[q], kinds[q + 1])            # dense and sparse wire: the same block in HBM
    # one value beyond 65 535, one non-integer: the block starts over as floats
    for bad in (65536.0, 2.5):
        Xb = X.copy(); Xb[7, 600] = bad
        sa.SHARP(sp.csc_matrix(Xb), ensize_K=2, rN_seed=11, logflag=False, prep=False)
        assert lib.sharp_x_wire() == 32
        sa.SHARP(np.asfortranarray(Xb), ensize_K=2, rN_seed=11, logflag=False, prep=False)
        assert lib.sharp_x_wire() == 32


def test_sparse_block_with_more_than_65536_genes_takes_int32_row_indices(sa, oracle):
    import torch
    from sharp_amd import device as dev

    m, n = 70001, 40
    rng = np.random.default_rng(5)
    X = np.zeros((m, n))
    for c in range(n):
        g = rng.choice(m, size=300, replace=False)
        X[g, c] = rng.integers(1, 30, size=300)
    X[m - 1, 0] = 9.0; X[65536, 1] = 4.0; X[65535, 2] = 3.0
    dX = dev.csc_to_dev(sp.csc_matrix(X))
    assert np.array_equal(dX.cpu().numpy(), X.T.astype(np.float32))
    pr = sa.Projector(m, 60, [2154])
    E = pr.project(X, logflag=True)                                       # (the dense host path of the same block: 8-bit values on the wire)
    assert sa.lib().sharp_x_wire() == 8
    refE = oracle.project(X, oracle.ranM(m, 60, 2154), True)
    np.testing.assert_allclose(E, refE, rtol=0, atol=2e-12 * np.abs(refE).max())
    del torch


def test_host_blocks_taken_in_groups_equal_block_after_block(sa, oracle, monkeypatch):
    """SHARP_unlimited on a list of host blocks (R/SHARP_unlimited.R:125-143): the blocks cross PCIe into a ring of resident copies and the
    compute thread takes those that arrived meanwhile TOGETHER as one pipelined batch (SHARP_HOST_GROUP, default 3); block after block
    (SHARP_HOST_GROUP=1) is the same call: labels and E1 rows identical, both equal to the oracle; dense and sparse lists alike."""
    sizes = [5200, 5100, 640, 5300, 5050]
    c0, blocks = 0, []
    for nb in sizes:
        blocks.append(oracle.synth_fill(77, 1400, c0, nb, 5, 140)); c0 += nb
    ref = oracle.SHARP_unlimited(blocks, K=3, rN_seed=2103, nthreads=8, want_view=True)
    out = {}
    for group in ("3", "1"):
        monkeypatch.setenv("SHARP_HOST_GROUP", group)
        for name, lst in (("dense", blocks), ("sparse", [sp.csc_matrix(b) for b in blocks])):
            res = sa.SHARP_unlimited(lst, ensize_K=3, rN_seed=2103)
            assert np.array_equal(res["pred_clusters"], ref["pred_clusters"]), (group, name)
            np.testing.assert_allclose(res["viE"], ref["viE"], rtol=0, atol=2e-12 * np.abs(ref["viE"]).max())
            out[(group, name)] = res["viE"]
    monkeypatch.delenv("SHARP_HOST_GROUP")
    for k, v in out.items():
        np.testing.assert_array_equal(v, out[("1", "dense")], err_msg=str(k))
