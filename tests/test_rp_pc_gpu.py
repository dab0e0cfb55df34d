"""The RP matmul as ONE producer / consumer kernel (rp3.hip, the default) against the library's other sparse form (rp.hip, the single
scatter kernel that takes unaligned blocks; SHARP_RP_KERNEL=fused) bit for bit -- both add the same fixed-point terms, in any order --
and against the oracle: R/RPmat.R:32, R/SHARP.R:343-345,569-585.  (The two-kernel form of rounds 2-3 is lab code: tools/lab/rp2.hip.)"""
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


def both(sa, monkeypatch, pr, X, logflag):
    monkeypatch.setenv("SHARP_RP_KERNEL", "fused")
    E2 = pr.project(X, logflag=logflag)
    monkeypatch.setenv("SHARP_RP_KERNEL", "pc")
    E3 = pr.project(X, logflag=logflag)
    monkeypatch.delenv("SHARP_RP_KERNEL")
    return E2, E3


@pytest.mark.parametrize("m,n,K,logflag", [(1500, 96, 3, True), (1500, 96, 3, False), (2003, 130, 1, True), (6000, 700, 15, True),
                                            (4097, 1033, 5, True), (20000, 600, 5, True), (27000, 300, 5, True), (1024, 64, 5, True)])
def test_pc_kernel_equals_scatter_form_and_oracle(sa, oracle, monkeypatch, m, n, K, logflag):
    X = oracle.synth_fill(SEED, m, 0, n, 4, max(1, m // 8))
    p = int(np.ceil(np.log2(max(n, 2)) / 0.04)) if m < 20000 else 474
    seeds = [50 + 2103 + k for k in range(1, K + 1)]
    pr = sa.Projector(m, p, seeds)
    E2, E3 = both(sa, monkeypatch, pr, X, logflag)
    assert np.array_equal(E2, E3)                     # the same integer sums
    for k in range(min(K, 2)):
        ref = oracle.project(X[:, :64], oracle.ranM(m, p, seeds[k]), logflag)
        np.testing.assert_allclose(E3[:64, k * p:(k + 1) * p], ref, rtol=0, atol=2e-12 * np.abs(ref).max())


@pytest.mark.parametrize("m,n,p,K", [(900, 40, 600, 15), (64, 24, 150, 15)])   # two launch groups; every gene in overflow segments
def test_pc_kernel_launch_groups_and_overflow_segments(sa, oracle, monkeypatch, m, n, p, K):
    X = oracle.synth_fill(SEED, m, 0, n, 3, max(1, m // 4))
    seeds = [50 + 2103 + k for k in range(1, K + 1)]
    pr = sa.Projector(m, p, seeds)
    E2, E3 = both(sa, monkeypatch, pr, X, True)
    assert np.array_equal(E2, E3)
    ref = oracle.project(X, oracle.ranM(m, p, seeds[0]), True)
    np.testing.assert_allclose(E3[:, :p], ref, rtol=0, atol=2e-12 * np.abs(ref).max())


def test_pc_kernel_value_kinds_and_list_overflow(sa, oracle, monkeypatch):
    """Table values and values outside the table side by side (counts of 256 and more, non-integers, values below one), cells
    whose non-zeros outnumber the LDS list (every gene non-zero at m = 20 000: the entries beyond it go through the scratch
    block), empty cells between them, an fp64 (TPM-like) block, and raw (non-log) values."""
    m, p = 20000, 120
    pr = sa.Projector(m, p, [2154, 2155, 2156])
    n = 70
    rng = np.random.default_rng(11)
    X = np.zeros((m, n))
    X[:, 1] = np.arange(m) % 7 + 1                                   # dense, all table values
    X[:, 3] = np.where(np.arange(m) % 2 == 0, np.arange(m) % 300, 0.5)  # dense, both kinds interleaved
    X[1020:1030, 5] = [255, 256, 257, 0.25, 1, 2, 65535, 65536, 3.5, 254]
    X[:, 7] = np.linspace(0.0, 1000.0, m).astype(np.float32)         # dense, nearly all outside the table
    for c in range(8, n):
        nz = rng.choice(m, size=int(rng.integers(1, 9000)), replace=False)
        X[nz, c] = rng.integers(1, 400, size=nz.size)
    X[:, 20] = 0
    X[:, 21] = 0
    for logflag in (True, False):
        E2, E3 = both(sa, monkeypatch, pr, X, logflag)
        assert np.array_equal(E2, E3)
        assert np.all(E3[0] == 0) and np.all(E3[20] == 0)
        ref = oracle.project(X, oracle.ranM(m, p, 2154), logflag)
        np.testing.assert_allclose(E3[:, :p], ref, rtol=0, atol=2e-12 * np.abs(ref).max())
    # non-fp32-exact doubles: the block is stored as fp64, and every entry of the producer / consumer kernel's lists has a 64-bit term slot in
    # LDS (round 5); dense cells overflow it into the scratch block, a cell of exact small counts takes the term table inside the fp64 block
    Xt = X / np.maximum(X.sum(0, keepdims=True), 1.0) * 1e6
    Xt[:, 1] = X[:, 1]
    Xt[:, 9] = np.where(np.arange(m) % 3 == 0, 2.0, 0.1234567891234)
    for logflag in (True, False):
        E2, E3 = both(sa, monkeypatch, pr, Xt, logflag)
        assert sa.lib().sharp_x_storage() == 64
        assert np.array_equal(E2, E3)
        ref = oracle.project(Xt, oracle.ranM(m, p, 2154), logflag)
        np.testing.assert_allclose(E3[:, :p], ref, rtol=0, atol=2e-12 * np.abs(ref).max())


def test_pc_kernel_resident_block_bit_identical_at_cfg3_shape(sa):
    import torch

    lib = sa.lib()
    m, n, K, p = 20000, 8192, 5, 474
    pr = sa.Projector(m, p, [50 + 2103 + k for k in range(1, K + 1)])
    dX = torch.empty((n, m), dtype=torch.float32, device="cuda")
    assert lib.sharp_synth_fill_dev(C.c_uint(SEED), m, C.c_longlong(0), n, 12, 1000, C.c_void_p(dX.data_ptr()), C.c_longlong(m)) == 0
    outs = {}
    import os
    for kern in ("fused", "pc", "pc"):
        os.environ["SHARP_RP_KERNEL"] = kern
        sa.reload_options()
        dE = torch.zeros((n, K * p), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        assert lib.sharp_project_dev(pr.handle, C.c_void_p(dX.data_ptr()), m, n, C.c_longlong(m), 1, C.c_void_p(dE.data_ptr()),
                                     C.c_longlong(K * p)) == 0
        assert lib.sharp_synchronize() == 0
        outs.setdefault(kern, []).append(dE)
    del os.environ["SHARP_RP_KERNEL"]
    sa.reload_options()
    assert torch.equal(outs["fused"][0], outs["pc"][0]) and torch.equal(outs["pc"][0], outs["pc"][1])


def test_dense_cells_and_non_table_values_are_bit_identical_across_kernel_forms(sa, oracle, monkeypatch):
    """Cells whose entries outnumber the LDS list, and cells that interleave table values with values outside the table: the producer /
    consumer kernel's general path gives the same integer sums as the two-kernel form.  (Round 4's count-class mode, for which this data
    was built, left the product source in round 5: tools/lab/rp3_cls_lab.hip.)"""
    m, p, K = 20000, 474, 5
    n = 300
    X = oracle.synth_fill(SEED, m, 0, n, 12, 1000)
    X[:, 3] = np.arange(m) % 5                                          # dense: beyond the LDS part of the list
    X[:, 4] = np.where(np.arange(m) % 2 == 0, 1.5, 7.25)                # dense: a table value and a non-table value interleaved
    X[100:140, 5] = np.linspace(0.1, 900.0, 40)
    seeds = [50 + 2103 + k for k in range(1, K + 1)]
    pr = sa.Projector(m, p, seeds)
    E1 = pr.project(X, logflag=True)
    E1raw = pr.project(X, logflag=False)
    monkeypatch.setenv("SHARP_RP_KERNEL", "fused")
    E2 = pr.project(X, logflag=True)
    monkeypatch.delenv("SHARP_RP_KERNEL")
    assert np.array_equal(E1, E2)
    ref = oracle.project(X[:, :8], oracle.ranM(m, p, seeds[0]), False)
    np.testing.assert_allclose(E1raw[:8, :p], ref, rtol=0, atol=2e-12 * np.abs(ref).max())
    refl = oracle.project(X[:, :8], oracle.ranM(m, p, seeds[0]), True)
    np.testing.assert_allclose(E1[:8, :p], refl, rtol=0, atol=2e-12 * np.abs(refl).max())
