"""GPU parity tests for rows a7-a12 (wMetaC, sMetaC, SHARP_small/large/unlimited) against the oracle."""
import os

import numpy as np
import pytest
from sklearn.metrics import adjusted_rand_score

pytestmark = pytest.mark.gpu

SEED = 20261003


@pytest.fixture(scope="module")
def sa():
    import sharp_amd

    sharp_amd.init(0)
    return sharp_amd


def _noisy_ensemble(rng, N, C, G, flip):
    base = rng.integers(1, G + 1, N)
    cols = []
    for c in range(C):
        col = (base + c) % G + 1          # renamed copy of the same partition
        f = rng.random(N) < flip
        col[f] = rng.integers(1, G + 1, f.sum())
        cols.append(col)
    return base, np.stack(cols, 1)


@pytest.mark.parametrize("N,C,G,flip", [(400, 5, 6, 0.05), (1200, 15, 9, 0.15), (300, 3, 4, 0.0),
                                        (2000, 5, 12, 0.02),    # a cfg3 fold: 2000 cells, K = 5, nearly consistent labelings (the saturating case)
                                        (2000, 15, 12, 0.3),    # a cfg2 fold with noisy labelings
                                        (97, 2, 3, 0.1)])       # two labelings only
def test_wmetac_stages_match_oracle(sa, oracle, N, C, G, flip):
    rng = np.random.default_rng(N + C)
    base, nC = _noisy_ensemble(rng, N, C, G, flip)
    ref = oracle.wMetaC(nC, sil_thre=0.35)
    res = sa.wMetaC(nC, sil_thre=0.35, debug=True)
    assert res["allC"] == ref["allC"]
    np.testing.assert_allclose(res["w1"], ref["w1"], rtol=1e-13)
    np.testing.assert_allclose(res["S"], ref["S"], rtol=1e-12, atol=1e-15)
    assert np.array_equal(res["S"] == 1.0, ref["S"] == 1.0)      # identical clusters -> exactly 1
    assert res["S"].max() <= 1.0
    assert np.array_equal(res["tf"], ref["tf"])
    assert np.array_equal(res["finalC"], ref["finalC"])
    np.testing.assert_allclose(res["x0"], ref["x0"], atol=1e-15)


def test_wmetac_string_labels_and_vote_tie_break(sa, oracle):
    # colour-name labels as the reference passes them; a 2-way tie goes to the id whose string sorts first
    names = np.array(["red", "purple", "blue", "yellow"])
    rng = np.random.default_rng(3)
    base = rng.integers(0, 4, 200)
    nC = np.stack([names[base], names[(base + 1) % 4], names[base], names[(base + 2) % 4]], 1)
    res = sa.wMetaC(nC, sil_thre=0.35)
    assert adjusted_rand_score(base, res["finalC"]) == 1.0


def test_smetac_matches_oracle(sa, oracle):
    rng = np.random.default_rng(11)
    n, p, G = 3000, 120, 5
    truth = rng.integers(0, G, n)
    centers = rng.normal(size=(G, p)) * 2
    E = centers[truth] + rng.normal(size=(n, p))
    # 4 folds, each with its own (over-segmented) cluster ids
    fold = np.repeat(np.arange(4), n // 4)
    sub = rng.integers(0, 2, n)
    labels = fold * 65536 + truth * 2 + sub + 1
    ref = oracle.sMetaC(labels, E)
    res = sa.sMetaC(labels, E)
    assert np.array_equal(res["tf"], ref["tf"])
    assert np.array_equal(res["finalColor"], ref["finalColor"])
    assert adjusted_rand_score(truth, res["finalColor"]) > 0.99


def test_sharp_small_matches_oracle(sa, oracle):
    m, n, G, nm = 2000, 300, 6, 200
    X = oracle.synth_fill(SEED, m, 0, n, G, nm)
    ref = oracle.SHARP(X, K=5, rN_seed=2103)
    res = sa.SHARP(X, ensize_K=5, rN_seed=2103, logflag=False, prep=False)
    assert res["path"] == "SHARP_small" and res["reduced.dim"] == ref["p"]
    assert np.array_equal(res["pred_clusters"], ref["pred_clusters"])      # identical integer labels
    np.testing.assert_allclose(res["viE"], ref["viE"], rtol=0, atol=2e-12 * np.abs(ref["viE"]).max())
    s = oracle.SHARP_small(X, K=5, rN_seed=2103)
    np.testing.assert_allclose(res["x0"], s["x0"], atol=1e-15)
    # the explicit SHARP_small entry point and the N.cluster-given variant
    r2 = sa.SHARP_small(X, ensize_K=5, rN_seed=2103)
    assert np.array_equal(r2["pred_clusters"], ref["pred_clusters"])


def test_sharp_small_with_the_dense_rp_form_gives_the_same_labels(sa, oracle, monkeypatch):
    # the dense-projector MFMA form of the RP matmul (rp_dense.hip) under the whole SHARP_small path: E differs from the sparse
    # kernels' in the last bits (fp64 FMA chains instead of exact fixed-point sums), the labels do not
    m, n, G, nm = 2000, 300, 6, 200
    X = oracle.synth_fill(SEED, m, 0, n, G, nm)
    ref = oracle.SHARP(X, K=5, rN_seed=2103)
    monkeypatch.setenv("SHARP_RP_KERNEL", "dense")
    res = sa.SHARP(X, ensize_K=5, rN_seed=2103, logflag=False, prep=False)
    monkeypatch.delenv("SHARP_RP_KERNEL")
    assert np.array_equal(res["pred_clusters"], ref["pred_clusters"])
    np.testing.assert_allclose(res["viE"], ref["viE"], rtol=0, atol=2e-12 * np.abs(ref["viE"]).max())


def test_sharp_large_matches_oracle(sa, oracle):
    # the SHARP_large path at reduced fold size: shuffle (set.seed(50); sample(n)), 5 folds with the last two
    # rebalanced, K*T tasks, per-fold wMetaC, cross-fold sMetaC, un-shuffle
    m, n, G, nm = 3000, 900, 6, 300
    X = oracle.synth_fill(SEED, m, 0, n, G, nm)
    truth = oracle.synth_cluster(SEED, range(n), G)
    ref = oracle.SHARP(X, K=5, base_ncells=300, partition_ncells=200, rN_seed=2103, nthreads=4)
    res = sa.SHARP(X, ensize_K=5, base_ncells=300, partition_ncells=200, rN_seed=2103, logflag=False, prep=False)
    assert res["path"] == "SHARP_large"
    ari = adjusted_rand_score(ref["pred_clusters"], res["pred_clusters"])
    assert ari >= 0.99, ari
    assert np.array_equal(res["pred_clusters"], ref["pred_clusters"])
    np.testing.assert_allclose(res["viE"], ref["viE"], rtol=0, atol=2e-12 * np.abs(ref["viE"]).max())
    # enresults$x0 (R/SHARP.R:717-731,761-779): the folds' wMetaC x0 block-diagonal, columns summed per sMetaC cluster, rows un-shuffled
    refL = oracle.SHARP_large(X, K=5, ng=200, rN_seed=2103, nthreads=4)
    assert np.array_equal(refL["pred_clusters"], ref["pred_clusters"])
    assert res["x0"].shape == refL["x0"].shape and np.all(res["x0"].max(1) >= 1.0)
    np.testing.assert_allclose(res["x0"], refL["x0"], rtol=0, atol=1e-15)
    assert adjusted_rand_score(truth, res["pred_clusters"]) > 0.5
    # one fold (T == 1, :738-746): x0 is the fold's own wMetaC x0, and every label becomes NA -> one cluster (quirk 2)
    res1 = sa.SHARP(X, ensize_K=5, base_ncells=300, partition_ncells=2000, rN_seed=2103, logflag=False, prep=False)
    ref1 = oracle.SHARP_large(X, K=5, ng=2000, rN_seed=2103, nthreads=4)
    assert res1["path"] == "SHARP_large" and np.array_equal(res1["pred_clusters"], ref1["pred_clusters"])
    assert res1["x0"].shape == ref1["x0"].shape
    np.testing.assert_allclose(res1["x0"], ref1["x0"], rtol=0, atol=1e-15)


@pytest.mark.parametrize("chunk", [4, 7, 25])
def test_sharp_large_pipelined_chunks_match_oracle(sa, oracle, monkeypatch, chunk):
    # the K*T base-clustering tasks cut into 7 / 4 / 1 chunks: with more than one chunk two are in flight at a time
    # (distance GEMM of chunk j + 1 beside the agglomeration of chunk j, statistics of chunk j beside the agglomeration of
    # chunk j + 1, on two streams and two workspace sets); SHARP_HC_PIPE=0 runs one chunk at a time.  Same labels either way.
    m, n, G, nm = 3000, 900, 6, 300
    X = oracle.synth_fill(SEED, m, 0, n, G, nm)
    ref = oracle.SHARP(X, K=5, base_ncells=300, partition_ncells=200, rN_seed=2103, nthreads=4)
    kw = dict(ensize_K=5, base_ncells=300, partition_ncells=200, rN_seed=2103, logflag=False, prep=False)
    monkeypatch.setenv("SHARP_HC_CHUNK", str(chunk))
    res = sa.SHARP(X, **kw)
    res_again = sa.SHARP(X, **kw)                        # the workspace sets are reused
    monkeypatch.setenv("SHARP_HC_PIPE", "0")
    res_serial = sa.SHARP(X, **kw)
    assert res["path"] == "SHARP_large"
    for r in (res, res_again, res_serial):
        assert np.array_equal(r["pred_clusters"], ref["pred_clusters"])
    np.testing.assert_array_equal(res["viE"], res_serial["viE"])
    np.testing.assert_array_equal(res["x0"], res_serial["x0"])


def test_sharp_large_with_two_launch_groups_of_projectors(sa, oracle):
    # K * reduced.ndim = 9000 components: two launch groups (13 + 2 projectors) of the RP kernel over the same block
    m, n, G, nm = 2000, 2400, 5, 200
    X = oracle.synth_fill(SEED, m, 0, n, G, nm)
    kw = dict(ensize_K=15, reduced_ndim=600, base_ncells=300, partition_ncells=1200, rN_seed=2103, logflag=False, prep=False)
    res = sa.SHARP(X, **kw)
    ref = oracle.SHARP(X, K=15, reduced_ndim=600, base_ncells=300, partition_ncells=1200, rN_seed=2103, nthreads=8, want_view=True)
    assert res["path"] == "SHARP_large" and res["reduced.dim"] == 600
    assert np.array_equal(res["pred_clusters"], ref["pred_clusters"])
    np.testing.assert_allclose(res["viE"], ref["viE"], rtol=0, atol=2e-12 * np.abs(ref["viE"]).max())


def test_sharp_unlimited_matches_oracle(sa, oracle):
    m, G, nm = 3000, 6, 300
    blocks = [oracle.synth_fill(SEED, m, i * 6000, 6000, G, nm) for i in range(2)]
    truth = oracle.synth_cluster(SEED, range(12000), G)
    ref = oracle.SHARP_unlimited(blocks, rN_seed=2103, nthreads=8, want_view=True)
    res = sa.SHARP_unlimited(blocks, rN_seed=2103)
    ari = adjusted_rand_score(ref["pred_clusters"], res["pred_clusters"])
    assert ari >= 0.99, ari
    assert np.array_equal(res["pred_clusters"], ref["pred_clusters"])
    # viewflag outputs (R/SHARP_unlimited.R:215-232): viE = E1 of every block, x0 = one-hot of the final labels
    assert res["reduced.dim"] == ref["p"] and res["viE"].shape == (12000, ref["p"])
    np.testing.assert_allclose(res["viE"], ref["viE"], rtol=0, atol=2e-12 * np.abs(ref["viE"]).max())
    x0 = res["x0"].toarray() if hasattr(res["x0"], "toarray") else res["x0"]
    assert x0.shape == (12000, res["N.pred_clusters"]) and np.array_equal(x0.argmax(1) + 1, res["pred_clusters"])
    assert "viE" not in sa.SHARP_unlimited(blocks, rN_seed=2103, viewflag=False)
    # ids ordered by decreasing cluster size (R/SHARP_unlimited.R:180-183)
    sizes = np.bincount(res["pred_clusters"])[1:]
    assert np.all(np.diff(sizes) <= 0)
    assert adjusted_rand_score(truth, res["pred_clusters"]) > 0.9


def test_sharp_unlimited2_matches_oracle(sa, oracle):
    """R/SHARP_unlimited2.R: log10, projections rounded to one decimal, one sMetaC over the fold-level clusters of all blocks."""
    m, G, nm = 3000, 6, 300
    blocks = [oracle.synth_fill(SEED, m, i * 6000, 6000, G, nm) for i in range(2)]
    truth = oracle.synth_cluster(SEED, range(12000), G)
    ref = oracle.SHARP_unlimited2(blocks, rN_seed=2103, nthreads=8, want_view=True)
    res = sa.SHARP_unlimited2(blocks, rN_seed=2103, logflag=False)
    assert np.array_equal(res["pred_clusters"], ref["pred_clusters"])
    assert res["reduced.ndim"] == ref["p"] and res["paras"]["logmark"] is True
    # E1 = mean over K of projections ROUNDED to one decimal: multiples of 0.1/K up to rounding; same values as the oracle
    # unless a projection sits within 1e-12 of a rounding boundary (then one entry moves by 0.1/K)
    d = np.abs(res["viE"] - ref["viE"])
    assert (d > 1e-9).mean() < 1e-6
    assert np.allclose(np.round(res["viE"] * 5 * 10), res["viE"] * 5 * 10, atol=1e-6)
    sizes = np.bincount(res["pred_clusters"])[1:]
    assert np.all(np.diff(sizes) <= 0)
    assert adjusted_rand_score(truth, res["pred_clusters"]) > 0.9
    # a different ensemble size / partition size / seed, and the one-decimal rounding rule itself
    small = [b[:, :3000] for b in blocks]
    ref2 = oracle.SHARP_unlimited2(small, K=3, partition_ncells=1500, rN_seed=7, nthreads=8)
    res2 = sa.SHARP_unlimited2(small, ensize_K=3, partition_ncells=1500, rN_seed=7, logflag=False, forview=False)
    assert res2["N.cells"] == 6000 and "viE" not in res2
    assert np.array_equal(res2["pred_clusters"], ref2["pred_clusters"])
    assert oracle.round1([0.25, 0.35, 0.15, 2.5, -0.25]).tolist() == [0.2, 0.3, 0.1, 2.5, -0.2]


def test_view_reduction_above_1e5_cells(sa, oracle):
    """R/SHARP_unlimited.R:217-225: viE = 1/sqrt(50) * E1 %*% ranM2(p, 50, seed) (the branch itself needs > 1e5 cells;
    the reduction is checked on a small E1).  E1 is not fp32-exact, so the block is stored as fp64 (upload.hip) and the
    projection meets the tolerance of every other one: 2e-12 of the largest entry."""
    from sharp_amd.api import _view_reduce

    rng = np.random.default_rng(7)
    E1 = rng.standard_normal((300, 420))
    K, seed = 5, 2103
    got = _view_reduce(E1, seed, K)
    ref = oracle.project(E1.T, oracle.ranM(420, 50, 50 + seed + K + 1), False)     # (1/sqrt(50)) * t(z0) %*% t(E1), cells x 50
    assert got.shape == (300, 50)
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-12 * np.abs(ref).max())


@pytest.mark.parametrize("sizes", [[5300, 640, 5050], [5300, 1650, 5050]])      # p = 336, and p = 339: an odd p goes through a padded copy
def test_view_reduction_on_the_device_per_block(sa, oracle, sizes):
    """The same reduction taken per block on the GPU (sharp_unlimited_view_dim / sharp_SHARP_unlimited_viewk_dev: what SHARP_unlimited does above
    1e5 cells, so that ncells x 50 doubles leave the GPU instead of ncells x p), forced here on a small list of ragged blocks: resident blocks in
    one call, the in-process multi-device entry, the per-block entry of a sharded rank -- against the oracle's E1 put through the oracle's
    projection, block by block."""
    import torch
    from sharp_amd import device as dev

    m, K, seed = 1400, 3, 2103
    host, c0 = [], 0
    for nb in sizes:
        host.append(oracle.synth_fill(77, m, c0, nb, 5, 140))
        c0 += nb
    n = sum(sizes)
    ref = oracle.SHARP_unlimited(host, K=K, rN_seed=seed, nthreads=8, want_view=True)
    p = ref["p"]
    z0 = oracle.ranM(p, 50, 50 + seed + K + 1)
    want = np.concatenate([oracle.project(ref["viE"][a:b].T, z0, False) for a, b in zip(np.cumsum([0] + sizes[:-1]), np.cumsum(sizes))])
    tol = 4e-12 * np.abs(want).max()
    blocks = [torch.from_numpy(np.ascontiguousarray(h.T.astype(np.float32))).cuda() for h in host]
    torch.cuda.synchronize()
    pred, npred, pu, viE = dev.unlimited_dev(blocks, ensize_K=K, rN_seed=seed, viewflag=True, view_dim=50)
    assert pu == p and viE.shape == (n, 50) and np.array_equal(pred, ref["pred_clusters"])
    np.testing.assert_allclose(viE, want, rtol=0, atol=tol)
    # not armed: the plain view (E1 itself), and the armed state does not leak into the next call
    pred2, _, _, viE2 = dev.unlimited_dev(blocks, ensize_K=K, rN_seed=seed, viewflag=True)
    assert viE2.shape == (n, p) and np.array_equal(pred2, pred)
    np.testing.assert_allclose(viE2, ref["viE"], rtol=0, atol=2e-12 * np.abs(ref["viE"]).max())
    # the host-list entries of the API arm it themselves above 1e5 cells only; armed by hand here through the one-shot switch
    from sharp_amd._lib import check, lib
    check(lib().sharp_unlimited_view_dim(50))
    pm, nm, _, vm = dev.unlimited_multi_dev(blocks, [0, 1, 0], [0, 0], ensize_K=K, rN_seed=seed, viewflag=True)
    assert np.array_equal(pm, pred)
    np.testing.assert_allclose(vm.reshape(-1)[: n * 50].reshape(n, 50), want, rtol=0, atol=tol)
    # a rank of the sharded run, block by block
    proj = sa.Projector(m, p, [50 + seed + k for k in range(1, K + 1)])
    at = 0
    for b, nb in enumerate(sizes):
        v = np.zeros((nb, 50))
        dev.unlimited_block_dev(blocks[b], p, proj.handle, K, seed, viE=v, view_dim=50)
        np.testing.assert_allclose(v, want[at:at + nb], rtol=0, atol=tol)
        at += nb
    proj.close()


def test_ARI_five_indices_match_oracle(sa, oracle):
    """R/ARI.R:20-42 -> clues::adjustedRand(label, res$pred_clusters): Rand, HA, MA, FM, Jaccard; labels may be strings."""
    rng = np.random.default_rng(21)
    for n, ga, gb in [(500, 4, 6), (3000, 12, 9), (50, 2, 2)]:
        a = rng.integers(1, ga + 1, n)
        b = np.where(rng.random(n) < 0.7, (a * 7) % gb + 1, rng.integers(1, gb + 1, n))
        ref = oracle.adjusted_rand(a, b)
        for got in (sa.ARI(a, b), sa.ARI(a, {"pred_clusters": b}),
                    sa.ARI(np.array(["type_%d" % v for v in a]), np.array(["c%d" % v for v in b]))):   # cell-type names vs cluster names
            assert sorted(got) == sorted(ref)
            for key in ref:
                assert abs(got[key] - ref[key]) < 1e-12, (key, got[key], ref[key])
        assert abs(ref["HA"] - adjusted_rand_score(a, b)) < 1e-12
    same = sa.ARI(a, a)
    assert all(abs(same[k] - 1.0) < 1e-15 for k in same)


def test_run_Mtimes_SHARP(sa, oracle):
    X = oracle.synth_fill(SEED, 1500, 0, 300, 4, 200)
    out = sa.run_Mtimes_SHARP(X, Mtimes=2, Kset=[3, 5], rN_seed=2103)
    assert sorted(out) == ["enSize_3", "enSize_5"] and sorted(out["enSize_3"]) == ["Run_1", "Run_2"]
    a, b = out["enSize_5"]["Run_1"], out["enSize_5"]["Run_2"]
    assert a["ensize.K"] == 5 and "viE" not in a
    assert np.array_equal(a["pred_clusters"], b["pred_clusters"])          # seeded runs repeat exactly


def test_reference_error_behaviour(sa, oracle):
    X = oracle.synth_fill(SEED, 500, 0, 50, 3, 100)
    with pytest.raises(sa.SharpError, match="rN.seed should be an integer"):
        sa.SHARP(X, rN_seed=1.5)
    with pytest.raises(sa.SharpError, match="numeric"):
        sa.SHARP(X, rN_seed="a")
    with pytest.raises(sa.SharpError, match="No expression data"):
        sa.SHARP(None)
    with pytest.raises(sa.SharpError, match="LIST"):
        sa.SHARP_unlimited("nope")


@pytest.mark.parametrize("trial", range(8))
def test_randomised_sharp_parity(sa, oracle, trial):
    """Randomised end-to-end sweep (tools/parity_sweep.py in miniature): random data seed, size class (SHARP_small,
    SHARP_large, SHARP_large + small-cluster merge), ensemble size, linkage and projector seed; labels bit-identical."""
    rng = np.random.default_rng(9000 + trial)
    seed = int(rng.integers(1, 2**31 - 1))
    lo, hi = [(300, 2000), (5200, 8000), (10001, 16000)][trial % 3]
    n, m, G = int(rng.integers(lo, hi)), int(rng.integers(1200, 2400)), int(rng.integers(3, 9))
    K = int(rng.choice([3, 5, 7]))
    hm = str(rng.choice(["ward.D", "average", "complete", "ward.D2"]))
    rs = int(rng.integers(1, 5000))
    X = oracle.synth_fill(seed, m, 0, n, G, max(50, m // (2 * G)))
    ref = oracle.SHARP(X, K=K, rN_seed=rs, hmethod=hm, nthreads=8)
    # logflag=False: no testlog (the reference samples its cells with an unseeded RNG), log2 on -- what the oracle runs
    res = sa.SHARP(X, ensize_K=K, rN_seed=rs, hmethod=hm, forview=False, logflag=False)
    assert np.array_equal(res["pred_clusters"], ref["pred_clusters"]), (seed, n, m, G, K, hm, rs)
