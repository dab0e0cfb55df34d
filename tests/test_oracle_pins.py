"""Pins the CPU oracle (oracle/sharp_oracle.c) against everything available in this
container: well-known R outputs for the RNG restatement,
scipy ward linkage (App. B), sklearn silhouette / ARI.  The reference itself has no
tests or golden vectors for this path -> "parity unpinned" vs real R."""
import numpy as np
import pytest
from scipy.cluster.hierarchy import fcluster, linkage
from scipy.spatial.distance import squareform
from sklearn.metrics import adjusted_rand_score, silhouette_samples


def test_r_set_seed_runif_known_answers(oracle):
    # set.seed(1); runif(3)  etc. -- the familiar R outputs
    np.testing.assert_allclose(oracle.runif(1, 3), [0.2655087, 0.3721239, 0.5728534], atol=5e-8)
    np.testing.assert_allclose(oracle.runif(42, 2), [0.9148060, 0.9370754], atol=5e-8)
    np.testing.assert_allclose(oracle.runif(123, 3), [0.2875775, 0.7883051, 0.4089769], atol=5e-8)


def test_r_sample_permutation_known_answers(oracle):
    # R >= 3.6 (sample.kind = "Rejection"): set.seed(s); sample(10)
    assert oracle.sample_perm(42, 10).tolist() == [1, 5, 10, 8, 2, 4, 6, 9, 7, 3]
    assert oracle.sample_perm(123, 10).tolist() == [3, 10, 2, 8, 6, 9, 1, 7, 5, 4]
    assert oracle.sample_perm(1, 10).tolist() == [9, 4, 7, 1, 2, 5, 3, 10, 6, 8]


def test_ranM_density_and_first_draws(oracle):
    m, p = 4000, 60
    t = oracle.ranM(m, p, 2154)
    s = np.sqrt(m)
    dens = (t != 0).mean()
    assert abs(dens - 1 / s) < 0.1 / s
    assert abs((t > 0).sum() - (t < 0).sum()) < 5 * np.sqrt(m * p / s)
    # element i = r*p + c consumes the i-th uniform: u <= P -> 0; u <= P+q -> -1; else +1
    u = oracle.runif(2154, 200)
    P = 1 - 1 / s
    q = 1 / (2 * s)
    exp = np.where(u <= P, 0, np.where(u <= P + q, -1, 1))
    assert t.ravel()[:200].tolist() == exp.tolist()


def _toy(n=60, p=20, seed=0):
    rng = np.random.default_rng(seed)
    Y = rng.normal(size=(n, p))
    Y[: n // 2] += 1.0
    return Y


def test_cor_dist_matches_numpy(oracle):
    Y = _toy()
    d = oracle.cor_dist(oracle.scale_rows(Y))
    ref = (1 - np.corrcoef(Y))[np.triu_indices(Y.shape[0], 1)]
    np.testing.assert_allclose(d, ref, atol=1e-14)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_ward_d_matches_scipy(oracle, seed):
    # hclust(d, "ward.D") == scipy ward (= ward.D2) on sqrt(d), heights squared
    Y = _toy(seed=seed)
    n = Y.shape[0]
    d = oracle.cor_dist(oracle.scale_rows(Y))
    ia, ib, crit = oracle.hclust(d, n, "ward.D")
    L = linkage(np.sqrt(d), "ward")
    np.testing.assert_allclose(np.sort(crit), np.sort(L[:, 2] ** 2), rtol=1e-11)
    r = oracle.get_opt_hclust(Y)
    for c, k in enumerate(range(2, 12)):
        assert adjusted_rand_score(fcluster(L, k, "maxclust"), r["v"][:, c]) == 1.0
        # cutree numbering: ids by first appearance in observation order
        lab = r["v"][:, c]
        first = [np.argmax(lab == j) for j in range(1, k + 1)]
        assert first == sorted(first)


@pytest.mark.parametrize("method,scipy_name", [("single", "single"), ("complete", "complete"), ("average", "average"),
                                               ("mcquitty", "weighted")])
def test_other_linkages_match_scipy(oracle, method, scipy_name):
    Y = _toy(seed=3)
    n = Y.shape[0]
    d = oracle.cor_dist(oracle.scale_rows(Y))
    _, _, crit = oracle.hclust(d, n, method)
    L = linkage(d, scipy_name)
    np.testing.assert_allclose(np.sort(crit), np.sort(L[:, 2]), rtol=1e-11)


def test_silhouette_matches_sklearn(oracle):
    Y = _toy(seed=4)
    d = oracle.cor_dist(oracle.scale_rows(Y))
    r = oracle.get_opt_hclust(Y)
    D = squareform(d)
    for c in range(6):
        cl = r["v"][:, c]
        np.testing.assert_allclose(oracle.silhouette_widths(cl, d), silhouette_samples(D, cl, metric="precomputed"),
                                   atol=1e-13)
        assert abs(np.median(oracle.silhouette_widths(cl, d)) - r["msil"][c]) < 1e-15


def test_adjusted_rand_matches_sklearn(oracle):
    rng = np.random.default_rng(5)
    a = rng.integers(1, 6, 500)
    b = np.where(rng.random(500) < 0.8, a, rng.integers(1, 6, 500))
    r = oracle.adjusted_rand(a, b)
    assert abs(r["HA"] - adjusted_rand_score(a, b)) < 1e-12
    assert 0 < r["Jaccard"] < r["FM"] < 1 and r["HA"] < r["Rand"]
    assert abs(r["MA"] - r["HA"]) < 0.01


def test_folds_follow_reference_rebalancing(oracle):
    # R/SHARP.R:513-536: last two folds are balanced to floor/ceil(nt/2)
    f, T = oracle.make_folds(50000, 2000)
    assert T == 25 and np.bincount(f)[1:].tolist() == [2000] * 25
    f, T = oracle.make_folds(5300, 2000)
    assert T == 3 and np.bincount(f)[1:].tolist() == [2000, 1650, 1650]
    f, T = oracle.make_folds(900, 200)
    assert T == 5 and np.bincount(f)[1:].tolist() == [200, 200, 200, 150, 150]
    assert np.all(np.diff(f) >= 0)


def test_wmetac_identical_partitions_give_unit_similarity(oracle):
    # identical clusters across RPs -> S == 1 exactly and the vote recovers them
    rng = np.random.default_rng(6)
    base = rng.integers(1, 5, 120)
    nC = np.stack([base, (base % 4) + 1, base], 1)  # renamed copies of one partition
    r = oracle.wMetaC(nC)
    S = r["S"]
    assert S.max() <= 1.0 and np.all(np.diag(S) == 1.0)
    assert np.sum(S == 1.0) == 4 * 9  # each of 4 clusters appears in 3 columns
    assert adjusted_rand_score(base, r["finalC"]) == 1.0
    assert r["x0"].shape == (120, 4) and np.all(r["x0"].max(1) == 1.0)


def test_sharp_small_recovers_planted_clusters(oracle):
    seed, m, n, G, nm = 20261003, 2000, 300, 6, 200
    X = oracle.synth_fill(seed, m, 0, n, G, nm)
    truth = oracle.synth_cluster(seed, range(n), G)
    r = oracle.SHARP_small(X, K=5, rN_seed=2103)
    assert r["rc"] == 0
    assert adjusted_rand_score(truth, r["pred_clusters"]) > 0.95
    # pred ids numbered by first appearance (R/SHARP.R:429-443)
    lab = r["pred_clusters"]
    first = [np.argmax(lab == j) for j in range(1, lab.max() + 1)]
    assert first == sorted(first)
