"""GPU parity tests for rows a1/a2 (projector + RP matmul) through the C ABI, against the oracle."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 20261003


@pytest.fixture(scope="module")
def sa():
    import sharp_amd

    sharp_amd.init(0)
    return sharp_amd


@pytest.mark.parametrize("m,p,seed", [(500, 40, 2154), (4000, 246, 2155), (20000, 391, 2168),
                                      (20000, 474, 2158),      # cfg3's projectors (50 + 2103 + 5)
                                      (27000, 508, 2154),      # cfg4's
                                      (20000, 582, 2156)])     # cfg5's
def test_projector_matches_r_stream(sa, oracle, m, p, seed):
    pr = sa.ranM2(m, p, seed)
    g, c, s = pr.triplets(0)
    t = oracle.ranM(m, p, seed)
    gi, ci = np.nonzero(t)
    assert g.tolist() == gi.tolist() and c.tolist() == ci.tolist()
    assert s.tolist() == t[gi, ci].tolist()
    assert pr.nnz() == gi.size


def test_projector_device_build_equals_host_build(sa, oracle, monkeypatch):
    """The GPU Mersenne-Twister build and the host build of the same projectors give the same lists and the same E."""
    m, p, n = 5000, 300, 40
    seeds = [50 + 2103 + k for k in range(1, 8)]
    X = oracle.synth_fill(SEED, m, 0, n, 4, 600)
    dev = sa.Projector(m, p, seeds)
    monkeypatch.setenv("SHARP_PROJ_HOST", "1")
    host = sa.Projector(m, p, seeds)
    monkeypatch.delenv("SHARP_PROJ_HOST")
    assert dev.nnz() == host.nnz()
    for k in range(len(seeds)):
        for a, b in zip(dev.triplets(k), host.triplets(k)):
            assert np.array_equal(a, b)
    assert np.array_equal(dev.project(X, True), host.project(X, True))


@pytest.mark.parametrize("m,n,K,logflag", [(1500, 96, 3, True), (1500, 96, 3, False), (2003, 130, 1, True),
                                            (6000, 70, 15, True), (4097, 33, 5, True),
                                            (17, 40, 2, True),         # the smallest block the sparse kernels take (m <= 16: the dense form)
                                            (1024, 64, 5, False),      # exactly one unit of genes, raw values
                                            (20000, 66, 5, True),      # cfg3's gene count: twenty units, the last one ragged
                                            (27000, 48, 5, True),      # cfg4's gene count
                                            (3001, 257, 7, True)])     # odd gene count, more cells than one pass of the grid
def test_rp_matmul_matches_oracle(sa, oracle, m, n, K, logflag):
    X = oracle.synth_fill(SEED, m, 0, n, 4, max(1, m // 8))
    p = int(np.ceil(np.log2(max(n, 2)) / 0.04))
    seeds = [50 + 2103 + k for k in range(1, K + 1)]
    pr = sa.Projector(m, p, seeds)
    E = pr.project(X, logflag=logflag)
    assert E.shape == (n, K * p)
    for k in range(K):
        ref = oracle.project(X, oracle.ranM(m, p, seeds[k]), logflag)
        # tolerance: fp64 oracle sums fl(v*x) sequentially; the kernel sums 2^-44 fixed point exactly
        np.testing.assert_allclose(E[:, k * p:(k + 1) * p], ref, rtol=0, atol=2e-12 * np.abs(ref).max())


@pytest.mark.parametrize("m,n,p,K", [(900, 40, 600, 15),     # K*p = 9000 > 8191 components: two launch groups (13 + 2 projectors)
                                     (64, 24, 150, 15)])     # sqrt(m) = 8: ~280 codes per gene, every gene runs over several overflow segments
def test_rp_launch_groups_and_overflow_segments(sa, oracle, m, n, p, K):
    # the packed row lists: 13-bit components per launch group, lanes of four same-sign codes, overflow segments behind the
    # fixed-stride ones (sharp_amd/csrc/projector.hpp); host and device builds of the pack must agree with the oracle's dense product
    X = oracle.synth_fill(SEED, m, 0, n, 3, max(1, m // 4))
    seeds = [50 + 2103 + k for k in range(1, K + 1)]
    pr = sa.Projector(m, p, seeds)
    E = pr.project(X, logflag=True)
    assert E.shape == (n, K * p)
    for k in range(K):
        ref = oracle.project(X, oracle.ranM(m, p, seeds[k]), True)
        np.testing.assert_allclose(E[:, k * p:(k + 1) * p], ref, rtol=0, atol=2e-12 * np.abs(ref).max())


def test_rp_edge_cases(sa, oracle):
    m, p = 777, 50
    pr = sa.Projector(m, p, [2154, 2155])
    # all-zero cells, a dense cell, a single cell, non-integer (TPM-like) values
    X = np.zeros((m, 5))
    X[:, 1] = np.arange(m) % 7 + 1
    X[:, 3] = np.linspace(0.0, 1000.0, m).astype(np.float32)
    E = pr.project(X, True)
    assert np.all(E[0] == 0) and np.all(E[2] == 0) and np.all(E[4] == 0)
    for k, sd in enumerate([2154, 2155]):
        ref = oracle.project(X, oracle.ranM(m, p, sd), True)
        np.testing.assert_allclose(E[:, k * p:(k + 1) * p], ref, rtol=0, atol=2e-12 * np.abs(ref).max())
    E1 = pr.project(X[:, 1:2], True)
    assert np.array_equal(E1[0], E[1])  # per-cell result independent of the batch it is in
    # the compacted entries: counts below 256 travel as (gene, count) and take their term from the table, anything else
    # (counts >= 256, non-integers) carries its own 64-bit term -- both kinds side by side in one cell, on a 1024-gene unit boundary
    m2 = 2048
    pr2 = sa.Projector(m2, p, [2154])
    Y = np.zeros((m2, 3))
    Y[:, 0] = (np.arange(m2) % 3 == 0) * (np.arange(m2) % 520)            # integers on both sides of 255 / 256
    Y[:, 1] = np.where(np.arange(m2) % 2 == 0, np.arange(m2) % 300, 0.5)  # table values and non-integers interleaved
    Y[1020:1030, 2] = [255, 256, 257, 0.25, 1, 2, 65535, 65536, 3.5, 254]
    E2 = pr2.project(Y, True)
    ref2 = oracle.project(Y, oracle.ranM(m2, p, 2154), True)
    np.testing.assert_allclose(E2, ref2, rtol=0, atol=2e-12 * np.abs(ref2).max())
    Ee = pr.project(np.zeros((m, 0)), True)
    assert Ee.shape == (0, 2 * p)


def test_rp_is_bit_reproducible_and_linear_in_blocks(sa, oracle):
    import torch

    lib = sa.lib()
    m, n, K = 20000, 4096, 5
    p = 474
    pr = sa.Projector(m, p, [50 + 2103 + k for k in range(1, K + 1)])
    dX = torch.empty((n, m), dtype=torch.float32, device="cuda")
    assert lib.sharp_synth_fill_dev(C.c_uint(SEED), m, C.c_longlong(0), n, 12, 1000, C.c_void_p(dX.data_ptr()),
                                    C.c_longlong(m)) == 0
    outs = []
    for _ in range(2):
        dE = torch.zeros((n, K * p), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        assert lib.sharp_project_dev(pr.handle, C.c_void_p(dX.data_ptr()), m, n, C.c_longlong(m), 1,
                                     C.c_void_p(dE.data_ptr()), C.c_longlong(K * p)) == 0
        assert lib.sharp_synchronize() == 0
        outs.append(dE)
    assert torch.equal(outs[0], outs[1])  # integer accumulation -> identical bits run to run
    # a sub-block projected alone gives the same rows (size-independent property at cfg-scale m, p)
    dE2 = torch.zeros((100, K * p), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    assert lib.sharp_project_dev(pr.handle, C.c_void_p(dX[1000:1100].data_ptr()), m, 100, C.c_longlong(m), 1,
                                 C.c_void_p(dE2.data_ptr()), C.c_longlong(K * p)) == 0
    lib.sharp_synchronize()
    assert torch.equal(dE2, outs[0][1000:1100])
    # spot-check 8 cells against the oracle
    Xh = dX[:8].cpu().numpy().T.astype(np.float64)
    assert np.array_equal(Xh, oracle.synth_fill(SEED, m, 0, 8, 12, 1000))  # generator identical on CPU and GPU
    ref = oracle.project(Xh, oracle.ranM(m, p, 2154), True)
    np.testing.assert_allclose(outs[0][:8, :p].cpu().numpy(), ref, rtol=0, atol=2e-12 * np.abs(ref).max())


@pytest.mark.parametrize("m,n,p,K,logflag,tpm", [(900, 40, 600, 15, True, False),    # two launch groups
                                                 (64, 24, 150, 15, True, False),     # every gene runs over several overflow segments
                                                 (1500, 200, 192, 5, True, True),    # an fp64 block (TPM-like values), several 64-cell tiles
                                                 (1500, 96, 165, 3, False, False)])  # raw values
def test_rp_dense_mfma_form_matches_oracle(sa, oracle, monkeypatch, m, n, p, K, logflag, tpm):
    """The dense-projector form of the RP matmul (rp_dense.hip: the projector scattered into a dense +-1 matrix, log2(1 + x) transposed,
    one f64 MFMA GEMM per launch group) against the oracle's product, and against the sparse kernels on the same input."""
    X = oracle.synth_fill(SEED, m, 0, n, 3, max(1, m // 4))
    if tpm:
        X = X / np.maximum(X.sum(0, keepdims=True), 1.0) * 1e6      # non-fp32-exact doubles: the block is stored as fp64
    seeds = [50 + 2103 + k for k in range(1, K + 1)]
    pr = sa.Projector(m, p, seeds)
    E_sparse = pr.project(X, logflag=logflag)
    monkeypatch.setenv("SHARP_RP_KERNEL", "dense")
    E = pr.project(X, logflag=logflag)
    monkeypatch.delenv("SHARP_RP_KERNEL")
    assert E.shape == (n, K * p)
    for k in range(K):
        ref = oracle.project(X, oracle.ranM(m, p, seeds[k]), logflag)
        np.testing.assert_allclose(E[:, k * p:(k + 1) * p], ref, rtol=0, atol=2e-12 * np.abs(ref).max())
    np.testing.assert_allclose(E, E_sparse, rtol=0, atol=2e-12 * np.abs(E_sparse).max())


def test_rp_non_sparse_projector_takes_the_dense_form(sa, oracle):
    """m <= 16 genes: density 1/sqrt(m) >= 1/4, the projector is not sparse and the library itself picks the dense form."""
    m, n, p = 12, 70, 40
    rng = np.random.default_rng(5)
    X = rng.integers(0, 9, size=(m, n)).astype(np.float64)
    pr = sa.Projector(m, p, [2154, 2155, 2156])
    E = pr.project(X, logflag=True)
    for k, sd in enumerate([2154, 2155, 2156]):
        ref = oracle.project(X, oracle.ranM(m, p, sd), True)
        np.testing.assert_allclose(E[:, k * p:(k + 1) * p], ref, rtol=0, atol=2e-12 * max(np.abs(ref).max(), 1.0))

