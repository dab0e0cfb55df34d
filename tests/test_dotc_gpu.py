"""The sharp_C_* entry points (include/sharp_hip.h, sharp_amd/csrc/dotc.hip) called the way R's .C() calls a native routine:
every argument a pointer into a caller-owned vector (double* / int* / char**), void return, the status in the last argument --
these are the calls r/sharp_hip.R makes.  Results are compared with the oracle, like the plain entry points' tests."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = 20261003


def I(*v):
    return np.array(v, np.int32)


def D(*v):
    return np.array(v, np.float64)


def P(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.fixture(scope="module")
def lib():
    import sharp_amd

    sharp_amd.init(0)
    L = sharp_amd.lib()
    for name in dir(L):
        pass
    return L


def dotc(lib, name, *args):
    """.C(name, ...): all-pointer call, returns nothing; the caller reads its own vectors afterwards"""
    fn = getattr(lib, name)
    fn.restype = None
    fn(*[P(a) if isinstance(a, np.ndarray) else a for a in args])


def last_error(lib):
    buf = C.create_string_buffer(b" " * 2047)
    ptr = (C.c_char_p * 1)(C.addressof(buf))
    ln = I(2048)
    lib.sharp_C_last_error.restype = None
    lib.sharp_C_last_error(ptr, P(ln))
    return buf.value.decode()


def test_dotc_sharp_small_with_allrpinfo(lib, oracle):
    m, n, K = 1500, 400, 3
    X = oracle.synth_fill(SEED, m, 0, n, 4, 200)                      # genes x cells, column-major: as.double(scExp)
    p = int(np.ceil(np.log2(n) / 0.04))
    pred, viE, x0 = np.zeros(n, np.int32), np.zeros(n * p), np.zeros(n * 42)
    info, st = np.zeros(5, np.int32), I(-1)
    dotc(lib, "sharp_C_SHARP", X, I(m), D(n), I(K), I(0), I(0), I(0), I(1), I(0), I(0), I(0), I(0), I(0), D(-1.0), D(0.0), I(0), I(1),
         I(0), D(2103.0), pred, viE, x0, I(42), info, I(3), st)
    assert st[0] in (0, 16, 32, 48), last_error(lib)
    ref = oracle.SHARP_small(X, K=K, rN_seed=2103)
    assert info[4] == 0 and info[2] == p and info[3] == K and info[0] == pred.max()
    assert np.array_equal(pred, ref["pred_clusters"])
    np.testing.assert_allclose(viE.reshape(n, p), ref["viE"], rtol=0, atol=2e-12 * np.abs(ref["viE"]).max())
    np.testing.assert_allclose(x0[: n * info[1]].reshape(info[1], n).T, ref["x0"], atol=1e-15)
    # allrpinfo (R/SHARP.R:350-387,446): rowColor per projection and the projected matrices
    dims, enrp, indE = np.zeros(3, np.int32), np.zeros(n * K, np.int32), np.zeros(n * K * p)
    dotc(lib, "sharp_C_last_rpinfo", dims, enrp, indE, I(3), st)
    assert st[0] == 0 and dims.tolist() == [n, K, p]
    assert np.array_equal(enrp.reshape(K, n).T, ref["enrp"])
    E = indE.reshape(n, K * p)
    np.testing.assert_allclose(E.reshape(n, K, p).mean(1), ref["viE"], rtol=0, atol=2e-12 * np.abs(ref["viE"]).max())
    for k in range(K):
        e_ref = oracle.project(X, oracle.ranM(m, p, 50 + 2103 + k + 1), True)
        np.testing.assert_allclose(E[:, k * p:(k + 1) * p], e_ref, rtol=0, atol=2e-12 * np.abs(e_ref).max())


def test_dotc_flashmark(lib, oracle):
    """R/get_opt_hclust.R:76-83: flashmark = TRUE is flashClust "ward" (= ward.D); with another method R stops at `||`."""
    X = oracle.synth_fill(SEED, 1200, 0, 300, 4, 150)
    n = 300
    out = {}
    for fm, hm in ((0, 1), (1, 1), (1, 4)):
        pred, info, st = np.zeros(n, np.int32), np.zeros(5, np.int32), I(-1)
        dotc(lib, "sharp_C_SHARP", X, I(1200), D(n), I(3), I(0), I(0), I(0), I(hm), I(0), I(0), I(0), I(0), I(0), D(-1.0), D(0.0), I(fm),
             I(1), I(0), D(2103.0), pred, D(0), D(0), I(0), info, I(0), st)
        out[(fm, hm)] = (int(st[0]), pred.copy())
    assert out[(0, 1)][0] == 0 and out[(1, 1)][0] == 0 and np.array_equal(out[(0, 1)][1], out[(1, 1)][1])
    assert out[(1, 4)][0] == 2 and "invalid 'y' type in 'x || y'" in last_error(lib)


def test_dotc_projector_and_project(lib, oracle):
    m, p, n = 900, 40, 64
    h, st = I(0), I(-1)
    dotc(lib, "sharp_C_projector_create", I(m), I(p), I(1), D(2154.0), h, st)
    assert st[0] == 0 and h[0] > 0
    nnz = D(0)
    dotc(lib, "sharp_C_projector_triplets", h, I(0), I(0), I(0), I(0), nnz, st)      # capacity 0: count only
    nn = int(nnz[0])
    g, c, s = np.zeros(nn, np.int32), np.zeros(nn, np.int32), np.zeros(nn, np.int32)
    nnz = D(nn)
    dotc(lib, "sharp_C_projector_triplets", h, I(0), g, c, s, nnz, st)
    assert st[0] == 0
    tern = oracle.ranM(m, p, 2154)
    R = np.zeros((m, p), np.int8)
    R[g, c] = s
    assert np.array_equal(R, tern)
    X = oracle.synth_fill(SEED, m, 0, n, 3, 100)
    E = np.zeros(n * p)
    dotc(lib, "sharp_C_project", h, X, I(m), I(n), I(1), E, st)
    ref = oracle.project(X, tern, True)
    np.testing.assert_allclose(E.reshape(n, p), ref, rtol=0, atol=2e-12 * np.abs(ref).max())
    dotc(lib, "sharp_C_projector_destroy", h, st)
    assert st[0] == 0
    dotc(lib, "sharp_C_project", h, X, I(m), I(n), I(1), E, st)                          # a stale handle is an error, not a crash
    assert st[0] != 0 and last_error(lib)


def test_dotc_get_opt_hclust_wmetac_smetac(lib, oracle):
    rng = np.random.default_rng(4)
    n, p, G = 500, 60, 5
    cen = rng.standard_normal((G, p))
    lab = rng.integers(0, G, n)
    E = cen[lab] + 0.4 * rng.standard_normal((n, p))
    nkmax = 39
    f, v, msil, ch = np.zeros(n, np.int32), np.zeros(n * nkmax, np.int32), np.zeros(nkmax), np.zeros(nkmax)
    maxsil, height, optN, nk, br, st = D(0), np.zeros(n - 1), I(0), I(0), I(0), I(-1)
    dotc(lib, "sharp_C_get_opt_hclust", np.ascontiguousarray(E), I(n), I(p), I(1), I(0), I(2), I(40), D(0.35), D(2.0), I(0), f, v, msil, ch,
         maxsil, height, optN, nk, br, I(7), st)
    ref = oracle.get_opt_hclust(E)
    assert st[0] == 0 and nk[0] == 39 and np.array_equal(f, ref["f"]) and np.array_equal(v.reshape(39, n).T, ref["v"])
    np.testing.assert_allclose(msil, ref["msil"], atol=1e-10)
    np.testing.assert_allclose(height, ref["height"], rtol=1e-9)
    rc_, ms = np.zeros(n, np.int32), D(0)
    dotc(lib, "sharp_C_getrowColor", np.ascontiguousarray(E), I(n), I(p), I(1), I(0), I(2), I(40), D(0.35), D(2.0), I(0), rc_, ms, st)
    assert st[0] == 0 and np.array_equal(rc_, oracle.getrowColor(E, height_Ntimes=2.0)["rowColor"])
    # wMetaC on an ensemble of renamed noisy copies
    C_ = 5
    nC = np.stack([((lab + c) % G + 1) for c in range(C_)], 1).astype(np.int32)
    nC[rng.random((n, C_)) < 0.1] = 1
    nCf = np.asfortranarray(nC)
    finalC, x0, ncl = np.zeros(n, np.int32), np.zeros(n * 42), I(0)
    dotc(lib, "sharp_C_wMetaC", nCf, I(n), I(C_), I(1), I(0), I(2), I(40), D(0.35), D(2.0), finalC, x0, ncl, I(1), st)
    refw = oracle.wMetaC(nC, sil_thre=0.35)
    assert st[0] in (0, 16) and np.array_equal(finalC, refw["finalC"])
    np.testing.assert_allclose(x0[: n * ncl[0]].reshape(ncl[0], n).T, refw["x0"], atol=1e-15)
    # sMetaC: n as double
    labels = (lab * 7 + rng.integers(0, 3, n)).astype(np.int32)
    fin, tf, nCo = np.zeros(n, np.int32), np.zeros(n, np.int32), I(0)
    dotc(lib, "sharp_C_sMetaC", labels, np.ascontiguousarray(E), D(n), I(p), I(1), I(0), I(2), I(40), D(0.35), D(2.0), fin, tf, nCo, st)
    refs = oracle.sMetaC(labels, E)
    assert st[0] in (0, 16) and nCo[0] == refs["nC"] and np.array_equal(fin, refs["finalColor"])


def test_dotc_unlimited_and_merge(lib, oracle):
    m, G, nm = 2000, 5, 250
    sizes = [5200, 5600]
    blocks = [oracle.synth_fill(SEED, m, o, s, G, nm) for o, s in zip([0, 5200], sizes)]
    Xcat = np.concatenate([b.ravel(order="F") for b in blocks])          # unlist(lapply(scExp, as.double))
    ncells = sum(sizes)
    p = int(np.ceil(np.log2(ncells) / 0.04))
    pred, viE, info, st = np.zeros(ncells, np.int32), np.zeros(ncells * p), np.zeros(2, np.int32), I(-1)
    dotc(lib, "sharp_C_SHARP_unlimited", Xcat, I(2), D(*sizes), I(m), I(3), I(0), I(0), I(0), D(2103.0), pred, viE, info, I(1), st)
    ref = oracle.SHARP_unlimited(blocks, K=3, rN_seed=2103, nthreads=8, want_view=True)
    assert st[0] in (0, 16, 32, 48) and info[1] == p == ref["p"] and info[0] == pred.max()
    assert np.array_equal(pred, ref["pred_clusters"])
    np.testing.assert_allclose(viE.reshape(ncells, p), ref["viE"], rtol=0, atol=2e-12 * np.abs(ref["viE"]).max())
    # the view dimension armed the way r/sharp_hip.R arms it above 1e5 cells: viE comes back ncells x 50, E1 reduced per block on the GPU
    v50 = np.zeros(ncells * 50)
    dotc(lib, "sharp_C_unlimited_view_dim", I(50), st)
    assert st[0] == 0
    dotc(lib, "sharp_C_SHARP_unlimited", Xcat, I(2), D(*sizes), I(m), I(3), I(0), I(0), I(0), D(2103.0), pred, v50, info, I(1), st)
    assert st[0] in (0, 16, 32, 48) and np.array_equal(pred, ref["pred_clusters"])
    z0 = oracle.ranM(p, 50, 50 + 2103 + 3 + 1)
    want = np.concatenate([oracle.project(ref["viE"][a:b].T, z0, False) for a, b in ((0, 5200), (5200, ncells))])
    np.testing.assert_allclose(v50.reshape(ncells, 50), want, rtol=0, atol=4e-12 * np.abs(want).max())
    rng = np.random.default_rng(2)
    M = rng.standard_normal((60, 16)) + np.repeat(rng.standard_normal((6, 16)) * 3, 10, 0)
    cnt = rng.integers(20, 900, 60).astype(np.float64)
    fid, nf = np.zeros(60, np.int32), I(0)
    dotc(lib, "sharp_C_unlimited_merge", np.ascontiguousarray(M), cnt, I(60), I(16), D(cnt.sum()), I(0), I(0), I(0), fid, nf, st)
    refm = oracle.unlimited_merge(M, cnt.astype(np.int64), int(cnt.sum()))
    assert st[0] in (0, 16) and nf[0] == refm["n_final"] and np.array_equal(fid, refm["final_id"])


def test_dotc_decision_log(lib, oracle):
    """sharp_C_decision_log / sharp_C_last_decisions as r/sharp_hip.R calls them (sharp_decision_log(), sharp_last_decisions()): the rows of a
    SHARP_small call against the oracle's log."""
    m, n, K = 1500, 400, 3
    X = oracle.synth_fill(SEED, m, 0, n, 4, 200)
    p = int(np.ceil(np.log2(n) / 0.04))
    pred, viE, x0 = np.zeros(n, np.int32), np.zeros(n * p), np.zeros(n * 42)
    info, st = np.zeros(5, np.int32), I(-1)
    dotc(lib, "sharp_C_decision_log", I(1), st)
    assert st[0] == 0
    dotc(lib, "sharp_C_SHARP", X, I(m), D(n), I(K), I(0), I(0), I(0), I(1), I(0), I(0), I(0), I(0), I(0), D(-1.0), D(0.0), I(0), I(1),
         I(0), D(2103.0), pred, viE, x0, I(42), info, I(3), st)
    assert st[0] in (0, 16, 32, 48), last_error(lib)
    rows, cnt = np.zeros(14 * 64), I(0)
    dotc(lib, "sharp_C_last_decisions", rows, I(64), cnt, st)
    assert st[0] == 0 and cnt[0] == K + 1
    got = rows[: 14 * cnt[0]].reshape(-1, 14)
    dotc(lib, "sharp_C_decision_log", I(0), st)
    dotc(lib, "sharp_C_last_decisions", rows, I(64), cnt, st)
    assert cnt[0] == 0
    oracle.decision_log(True)
    try:
        ref = oracle.SHARP(X, K=K, rN_seed=2103)
        want = oracle.last_decisions()
    finally:
        oracle.decision_log(False)
    assert np.array_equal(pred, ref["pred_clusters"])
    exact = [0, 1, 2, 3, 4, 5, 6, 7, 12, 13]
    assert got.shape == want.shape and np.array_equal(got[:, exact], want[:, exact])
    np.testing.assert_allclose(got[:, 8], want[:, 8], rtol=1e-8, atol=1e-10)
