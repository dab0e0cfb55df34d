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
