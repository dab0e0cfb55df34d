"""GPU parity test for SURVEY.md 8(f4): get_marker_genes' per-gene statistics against the oracle and against scipy."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 20261003


@pytest.fixture(scope="module")
def sa():
    import sharp_amd

    sharp_amd.init(0)
    return sharp_amd


def test_marker_gene_statistics_match_oracle_and_scipy(sa, oracle):
    from scipy import stats

    m, n, G = 1200, 900, 5
    X = oracle.synth_fill(SEED, m, 0, n, G, 200)                        # counts: long tie groups
    X[7] = 0.0                                                           # an all-zero gene: sparsity 0 -> (0, 0, 1, 0, 0)
    X[8, :3] = [2.5, 0.5, 0.5]                                           # a very sparse gene (3 of 900 cells)
    rng = np.random.default_rng(3)
    X[9] = rng.gamma(2.0, 1.0, n).astype(np.float32)                     # dense, non-integer (fp32-exact: the block format is fp32)
    X[10] = np.where(rng.random(n) < 0.5, -rng.random(n), rng.random(n)).astype(np.float32)   # negatives rank below the zeros
    X[10, ::7] = 0.0
    truth = oracle.synth_cluster(SEED, range(n), G) + 1
    for ng in (1, 3):
        ref = oracle.marker_genes(X, truth, G, theta=1e-4, ng=ng)
        res = sa.get_marker_genes(X, {"pred_clusters": truth}, ng=ng, pvalue=2.0, auc=-1.0, FC=-1.0)   # keep every gene
        got = np.zeros((m, 5))
        all_out = np.zeros((m, 5))
        import ctypes as C
        lab = truth.astype(np.int32)
        Xf = np.asfortranarray(X)
        sa.lib().sharp_marker_genes(Xf.ctypes.data_as(C.POINTER(C.c_double)), m, C.c_longlong(n), C.c_longlong(m),
                                    lab.ctypes.data_as(C.POINTER(C.c_int)), G, C.c_double(1e-4), ng,
                                    all_out.ctypes.data_as(C.POINTER(C.c_double)))
        assert np.array_equal(all_out[:, 1], ref[:, 1])                  # same cluster picked for every gene
        np.testing.assert_allclose(all_out[:, 0], ref[:, 0], rtol=1e-12, atol=1e-14)        # auc
        np.testing.assert_allclose(all_out[:, 3], ref[:, 3], rtol=0, atol=0)                # sparsity
        np.testing.assert_allclose(all_out[:, 4], ref[:, 4], rtol=1e-12)                    # FC
        ok = ref[:, 2] > 1e-290
        np.testing.assert_allclose(all_out[ok, 2], ref[ok, 2], rtol=1e-9)                   # p-value (erfc of a large z)
        assert all_out[7].tolist() == [0.0, 0.0, 1.0, 0.0, 0.0]
        del got, res
    # independent check of the statistics themselves on a few genes: scipy's Mann-Whitney (same normal approximation)
    ref = oracle.marker_genes(X, truth, G)
    for g in (0, 9, 10, 500):
        c = int(ref[g, 1])
        a, b = X[g, truth == c], X[g, truth != c]
        u = stats.mannwhitneyu(a, b, alternative="two-sided", method="asymptotic", use_continuity=True)
        assert abs(u.statistic / (a.size * b.size) - ref[g, 0]) < 1e-12
        assert abs(u.pvalue - ref[g, 2]) <= 1e-9 * max(u.pvalue, 1e-300)
    # the host part: Holm adjustment, thresholds, ordering
    res = sa.get_marker_genes(X, {"pred_clusters": truth}, gene_names=["g%d" % i for i in range(m)])
    mg = res["mginfo"]
    assert mg["gene"].size > 0 and np.all(mg["FC"] >= 2) and np.all(mg["pvalue"] < 0.01)
    assert np.all(np.diff(mg["icluster"]) >= 0)
    assert res["mat"].shape == (mg["gene"].size, n)
    p = np.array([0.01, 0.04, 0.03, 0.005])
    from sharp_amd.api import _p_adjust_holm
    assert np.allclose(_p_adjust_holm(p), [0.03, 0.06, 0.06, 0.02])      # p.adjust(c(.01,.04,.03,.005), "holm")


def test_marker_genes_over_block_lists_match_the_whole_matrix(sa, oracle, tmp_path):
    """get_marker_genes_unlimited (R/get_marker_genes_unlimited.R:95-146: the per-gene pass over the cells of ALL blocks of the list
    SHARP_unlimited clustered, ng = 1) on ragged dense and sparse blocks, and get_marker_genes_unlimited2
    (R/get_marker_genes_unlimited2.R:139-214: gene-wise partition files, ng = min(10, N.cluster)) against the oracle's per-gene pass
    on the whole matrix."""
    import ctypes as C

    import scipy.sparse as sp
    from sharp_amd import blocks as sblocks
    from sharp_amd.api import _p_adjust_holm

    m, G = 900, 5
    sizes = [400, 650, 300, 500]
    n = sum(sizes)
    X = oracle.synth_fill(SEED, m, 0, n, G, 150)
    X[7] = 0.0                                                           # zero in every block: dropped (:44-57)
    X[8, :3] = [2.5, 0.5, 0.5]
    rng = np.random.default_rng(5)
    X[9] = rng.gamma(2.0, 1.0, n).astype(np.float32)
    truth = oracle.synth_cluster(SEED, range(n), G) + 1
    cuts = np.cumsum([0] + sizes)
    dense = [X[:, cuts[b]:cuts[b + 1]] for b in range(len(sizes))]
    ref1 = oracle.marker_genes(X, truth, G, theta=1e-5, ng=1)
    # the C entries directly: dense resident blocks and dgCMatrix-like blocks give the oracle's rows for every gene
    lab = truth.astype(np.int32)
    ncb = np.array(sizes, np.int64)
    cs = [sp.csc_matrix(b) for b in dense]
    cps = [c.indptr.astype(np.int32) for c in cs]
    ris = [c.indices.astype(np.int32) for c in cs]
    vxs = [c.data.astype(np.float64) for c in cs]
    B = len(sizes)
    out = np.zeros((m, 5))
    rc = sa.lib().sharp_marker_genes_blocks_csc((C.POINTER(C.c_int) * B)(*[a.ctypes.data_as(C.POINTER(C.c_int)) for a in cps]),
                                                (C.POINTER(C.c_int) * B)(*[a.ctypes.data_as(C.POINTER(C.c_int)) for a in ris]),
                                                (C.POINTER(C.c_double) * B)(*[a.ctypes.data_as(C.POINTER(C.c_double)) for a in vxs]),
                                                ncb.ctypes.data_as(C.POINTER(C.c_longlong)), B, m, lab.ctypes.data_as(C.POINTER(C.c_int)), G,
                                                C.c_double(1e-5), 1, out.ctypes.data_as(C.POINTER(C.c_double)))
    assert rc == 0, sa.lib().sharp_last_error()
    assert np.array_equal(out[:, 1], ref1[:, 1])
    np.testing.assert_allclose(out[:, 0], ref1[:, 0], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(out[:, 3], ref1[:, 3], rtol=0, atol=0)
    ok = ref1[:, 2] > 1e-290
    np.testing.assert_allclose(out[ok, 2], ref1[ok, 2], rtol=1e-9)
    # the front ends
    y = {"pred_clusters": truth}
    names = np.array(["g%d" % i for i in range(m)])
    rd = sa.get_marker_genes_unlimited(dense, y, gene_names=names)
    rs = sa.get_marker_genes_unlimited([sp.csr_matrix(b) for b in dense], y, gene_names=names)
    keep = ref1[:, 3] > 0                                                # (all-zero genes are not rows at all)
    sel = keep & (ref1[:, 3] > 1e-5) & ~np.isnan(ref1[:, 2])
    padj = _p_adjust_holm(ref1[sel, 2])
    adauc = min(0.85, min(ref1[sel, 0][ref1[sel, 1] == c].max() for c in np.unique(ref1[sel, 1])))
    want = names[sel][(padj < 0.01) & (ref1[sel, 0] > adauc)]
    for r in (rd, rs):
        assert r["mginfo"]["gene"].tolist() == want.tolist() and want.size > 0
        assert r["mat"].shape == (want.size, n) and np.array_equal(r["mat"], X[[int(g[1:]) for g in want]])
        assert np.array_equal(r["label"], truth)
    assert "g7" not in rd["mginfo"]["gene"].tolist()
    # gene-wise partition files (unlimited2): three files of 300 genes, all cells each
    d = tmp_path / "genes"
    d.mkdir()
    for i in range(3):
        sblocks.write_block(str(d / ("part%d.blk" % (i + 1))), X[300 * i:300 * (i + 1)])
    assert np.array_equal(sblocks.read_block(str(d / "part2.blk")), X[300:600].astype(np.float32))
    r2 = sa.get_marker_genes_unlimited2(str(d), y)
    ref10 = oracle.marker_genes(X, truth, G, theta=1e-5, ng=min(10, G))
    sel = (ref10[:, 3] > 1e-5) & ~np.isnan(ref10[:, 2])
    padj = _p_adjust_holm(ref10[sel, 2])
    adauc = min(0.85, min(ref10[sel, 0][ref10[sel, 1] == c].max() for c in np.unique(ref10[sel, 1])))
    idx = np.nonzero(sel)[0][(padj < 0.05) & (ref10[sel, 0] > adauc)]
    assert r2["mginfo"]["gene"].tolist() == ["part%d.blk:%d" % (g // 300 + 1, g % 300) for g in idx] and idx.size > 0
    np.testing.assert_allclose(r2["mginfo"]["auc"], ref10[idx, 0], rtol=1e-12)
    assert np.array_equal(r2["mginfo"]["icluster"], ref10[idx, 1].astype(np.int64))
    assert r2["gallinfo"]["gene"].size == int(sel.sum())
