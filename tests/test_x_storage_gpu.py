"""Storage of the expression block in HBM (sharp_amd/csrc/upload.hip): fp32 when every value survives the round trip through
float, fp64 otherwise -- the reference computes log2(X + 1) and the projection in double (R/SHARP.R:110-117,343-345), and its own
example data are TPM (README.md:88,114), i.e. doubles that fp32 would perturb by 6e-8 relative.  The oracle always gets the fp64
values themselves."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = 20261003


@pytest.fixture(scope="module")
def sa():
    import sharp_amd

    sharp_amd.init(0)
    return sharp_amd


def _tpm(oracle, seed, m, n, G, nm, cell0=0, drop=True):
    """counts -> counts / column sum * 1e6, unrounded (what SHARP() computes for exp.type = "count", R/SHARP.R:110-114)"""
    X = oracle.synth_fill(seed, m, cell0, n, G, nm)
    if drop:
        X = X[X.sum(1) != 0]
    return np.asfortranarray(X / X.sum(0, keepdims=True) * 1e6)


def test_projection_of_tpm_like_doubles(sa, oracle):
    m, n, p = 2500, 96, 60
    X = _tpm(oracle, SEED, m, n, 4, 300)
    assert not np.array_equal(X.astype(np.float32).astype(np.float64), X)
    pr = sa.Projector(X.shape[0], p, [2154, 2155])
    for logflag in (True, False):
        E = pr.project(X, logflag=logflag)
        assert sa.lib().sharp_x_storage() == 64
        for k in range(2):
            ref = oracle.project(X, oracle.ranM(X.shape[0], p, 2154 + k), logflag)
            np.testing.assert_allclose(E[:, k * p:(k + 1) * p], ref, rtol=0, atol=2e-12 * np.abs(ref).max())
    # fp32-exact data stay fp32; integers beyond 2^24 and values beyond FLT_MAX are not exact
    Xi = oracle.synth_fill(SEED, m, 0, n, 4, 300)
    pr2 = sa.Projector(m, p, [2154])
    pr2.project(Xi)
    assert sa.lib().sharp_x_storage() == 32
    Xb = Xi.copy()
    Xb[3, 5] = 2.0**24 + 1
    Eb = pr2.project(Xb)
    assert sa.lib().sharp_x_storage() == 64
    refb = oracle.project(Xb, oracle.ranM(m, p, 2154), True)
    np.testing.assert_allclose(Eb, refb, rtol=0, atol=2e-12 * np.abs(refb).max())
    Xh = Xi.copy()
    Xh[7, 9] = 1e300                                           # log2(1 + 1e300) ~ 996.6: beyond fp32, fine in fp64
    Eh = pr2.project(Xh)
    refh = oracle.project(Xh, oracle.ranM(m, p, 2154), True)
    np.testing.assert_allclose(Eh, refh, rtol=0, atol=2e-12 * np.abs(refh).max())


def test_forced_fp64_storage_is_bit_identical_on_fp32_exact_data(sa, oracle, monkeypatch):
    X = oracle.synth_fill(SEED, 2000, 0, 5600, 5, 250)         # SHARP_large, counts
    a = sa.SHARP(X, ensize_K=3, rN_seed=2103, logflag=False)
    assert sa.lib().sharp_x_storage() == 32
    monkeypatch.setenv("SHARP_X_STORAGE", "fp64")
    b = sa.SHARP(X, ensize_K=3, rN_seed=2103, logflag=False)
    assert sa.lib().sharp_x_storage() == 64
    assert np.array_equal(a["pred_clusters"], b["pred_clusters"])
    assert np.array_equal(a["viE"], b["viE"])                  # same 44-bit fixed-point terms, integer sums: bit for bit


@pytest.mark.parametrize("trial", range(8))       # every kind twice, weak structure in two of them; the long sweep: tools/parity_sweep.py, profiles/r0*_parity_sweep.txt
def test_tpm_like_input_matches_oracle(sa, oracle, trial, monkeypatch):
    """The sweep of tools/parity_sweep.py on TPM-like doubles: SHARP_small, SHARP_large and SHARP_unlimited, the oracle on the same
    fp64 values; labels identical.  With the block forced to fp32 (the previous behaviour) the labels may differ: reported."""
    rng = np.random.default_rng(4200 + trial)
    seed = int(rng.integers(1, 2**31 - 1))
    kind = trial % 4
    K = int(rng.choice([3, 5]))
    rs = int(rng.integers(1, 5000))
    G = int(rng.integers(3, 8))
    m = int(rng.integers(1200, 2200))
    weak = trial % 3 == 2                                       # few marker genes: flat silhouette profiles, CH branch
    nm = int(rng.integers(15, 60)) if weak else max(50, m // (2 * G))
    if kind == 3:
        sizes = [int(rng.integers(5100, 6500)), int(rng.integers(300, 2500))]
        blocks, c0 = [], 0
        for s_ in sizes:
            blocks.append(_tpm(oracle, seed, m, s_, G, nm, c0, drop=False))   # (the same genes in every block)
            c0 += s_
        ref = oracle.SHARP_unlimited(blocks, K=K, rN_seed=rs, nthreads=8)["pred_clusters"]
        res = sa.SHARP_unlimited(blocks, ensize_K=K, rN_seed=rs, viewflag=False)["pred_clusters"]
        assert sa.lib().sharp_x_storage() == 64
        monkeypatch.setenv("SHARP_X_STORAGE", "fp32")
        r32 = sa.SHARP_unlimited(blocks, ensize_K=K, rN_seed=rs, viewflag=False)["pred_clusters"]
    else:
        n = int(rng.integers(*[(300, 2500), (5100, 8000), (10001, 14000)][kind]))
        X = _tpm(oracle, seed, m, n, G, nm)
        ref = oracle.SHARP(X, K=K, rN_seed=rs, nthreads=8, want_view=False)["pred_clusters"]
        res = sa.SHARP(X, ensize_K=K, rN_seed=rs, forview=False, logflag=False, prep=False)["pred_clusters"]
        assert sa.lib().sharp_x_storage() == 64
        monkeypatch.setenv("SHARP_X_STORAGE", "fp32")
        r32 = sa.SHARP(X, ensize_K=K, rN_seed=rs, forview=False, logflag=False, prep=False)["pred_clusters"]
    print("trial %d kind %d weak %d: fp32-narrowed block %s the fp64 labels" % (trial, kind, weak, "keeps" if np.array_equal(r32, ref) else "CHANGES"))
    assert np.array_equal(res, ref), (trial, seed, kind, K, rs)


def test_sparse_input_with_non_fp32_values(sa, oracle):
    import scipy.sparse as sp

    X = _tpm(oracle, SEED, 1800, 5300, 5, 200)
    ref = oracle.SHARP(X, K=3, rN_seed=11, nthreads=8, want_view=False)["pred_clusters"]
    dense = sa.SHARP(X, ensize_K=3, rN_seed=11, forview=False, logflag=False, prep=False)["pred_clusters"]
    sparse = sa.SHARP(sp.csc_matrix(X), ensize_K=3, rN_seed=11, forview=False, logflag=False, prep=False)["pred_clusters"]
    assert sa.lib().sharp_x_storage() == 64
    assert np.array_equal(dense, ref) and np.array_equal(sparse, ref)


def test_tpm_block_already_resident_as_fp64(sa, oracle):
    """The *_dev64 entry points: TPM-like doubles that are ALREADY in HBM (a torch float64 tensor) take the fp64 path without going
    back through the host -- same labels as the oracle on the same doubles, same projection as the host entry."""
    import ctypes as C

    import torch

    from sharp_amd import device as dev

    m0, n = 2500, 5400                                         # SHARP_large
    X = _tpm(oracle, SEED + 5, m0, n, 5, 300)
    m = X.shape[0]
    dX = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()    # (cells, genes) float64
    if dX.stride(0) % 2:                                       # even leading dimension (16-byte aligned columns)
        pad = torch.zeros((n, m + 1), dtype=torch.float64, device="cuda")
        pad[:, :m] = dX
        dX = pad[:, :m]
    torch.cuda.synchronize()
    ref = oracle.SHARP(X, K=3, rN_seed=2103, nthreads=8)
    pred, info = dev.SHARP_dev(dX, ensize_K=3, rN_seed=2103, flag=True)
    assert info["path"] == "SHARP_large" and np.array_equal(pred, ref["pred_clusters"])
    host = sa.SHARP(X, ensize_K=3, rN_seed=2103, logflag=True, prep=False)
    assert np.array_equal(pred, host["pred_clusters"])
    # the projection alone: bit-identical to the host entry's fp64 block, oracle tolerance
    p = 80
    pr = sa.Projector(m, p, [2154, 2155])
    dE = torch.zeros((n, 2 * p), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    lib = sa.lib()
    rc = lib.sharp_project_dev64(pr.handle, C.c_void_p(dX.data_ptr()), m, n, C.c_longlong(dX.stride(0)), 1, C.c_void_p(dE.data_ptr()),
                                 C.c_longlong(2 * p))
    assert rc == 0, lib.sharp_last_error()
    lib.sharp_synchronize()
    E = dE.cpu().numpy()
    np.testing.assert_array_equal(E, pr.project(X, logflag=True))
    refE = oracle.project(X, oracle.ranM(m, p, 2154), True)
    np.testing.assert_allclose(E[:, :p], refE, rtol=0, atol=2e-12 * np.abs(refE).max())
    # what the entry refuses: an odd leading dimension, a NaN
    bad = torch.zeros((8, 101), dtype=torch.float64, device="cuda")
    assert lib.sharp_project_dev64(pr.handle, C.c_void_p(bad.data_ptr()), 101, 8, C.c_longlong(101), 1, C.c_void_p(dE.data_ptr()),
                                   C.c_longlong(2 * p)) != 0
    dX[3, 7] = float("nan")
    torch.cuda.synchronize()
    assert lib.sharp_project_dev64(pr.handle, C.c_void_p(dX.data_ptr()), m, n, C.c_longlong(dX.stride(0)), 1, C.c_void_p(dE.data_ptr()),
                                   C.c_longlong(2 * p)) != 0
    assert "NaN" in lib.sharp_last_error().decode()
    pr.close()
