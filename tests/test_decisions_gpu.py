"""The decision log: every choice of a number of clusters on the path -- the middle one of the exact ties of the
median silhouette (R/get_opt_hclust.R:162-168), which.max(CHind) when max(msil) <= sil.thre (:194-195), the height-gap rule (:196-210),
sMetaC's two-cluster override (R/sMetaC.R:139-148) -- leaves one row on the GPU side (sharp_decision_log / sharp_last_decisions) and the
same row on the oracle's side.  The two logs are compared decision for decision: same call, same rule, same number of clusters, the same
number of exact ties, deciding values within the stage tolerances; the smallest margins per level are printed (pytest -s)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = 20261003
F = {n: i for i, n in enumerate(("level", "block", "k", "fold", "n", "branch", "chosen_k", "ties", "best", "runner_up", "sil_minus_thre",
                                 "height_ratio", "smetac_override_k", "levels"))}


@pytest.fixture(scope="module")
def sa():
    import sharp_amd

    sharp_amd.init(0)
    return sharp_amd


def compare_logs(sa, got, ref, what=""):
    """the GPU's rows against the oracle's: identical keys, rules, chosen numbers of clusters, tie counts and overrides; values close"""
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    exact = [F[c] for c in ("level", "block", "k", "fold", "n", "branch", "chosen_k", "ties", "smetac_override_k", "levels")]
    bad = np.flatnonzero((got[:, exact] != ref[:, exact]).any(1))
    assert bad.size == 0, (what, "first differing decision", got[bad[0]].tolist(), ref[bad[0]].tolist())
    for c, tol in (("best", 1e-9), ("runner_up", 1e-9), ("sil_minus_thre", 1e-9), ("height_ratio", 1e-8)):
        a, b = got[:, F[c]], ref[:, F[c]]
        assert np.array_equal(np.isnan(a), np.isnan(b)), (what, c)
        ok = ~np.isnan(a)
        ch = ok & (got[:, F["branch"]] >= 1) & (c in ("best", "runner_up"))        # CH values: relative
        np.testing.assert_allclose(a[ok & ~ch], b[ok & ~ch], rtol=0, atol=tol, err_msg=what + " " + c)
        np.testing.assert_allclose(a[ch], b[ch], rtol=1e-8, err_msg=what + " CH " + c)
    m = sa.decision_margins(got)
    print("\n%s decision margins per level: %s" % (what, m))
    return m


def _data(oracle, m, n, G, nm, cell0=0):
    return oracle.synth_fill(SEED, m, cell0, n, G, nm)


@pytest.mark.parametrize("nm,K", [(300, 4), (25, 3)])          # strong markers: the silhouette rule; 25 marker genes: CH / height-gap decisions
def test_small_and_large_path_logs_equal_the_oracles(sa, oracle, nm, K):
    X = _data(oracle, 2500, 700, 5, nm)
    for kw_gpu, kw_or in ((dict(), dict()), (dict(base_ncells=100, partition_ncells=200), dict(base_ncells=100, partition_ncells=200))):
        sa.decision_log(True); oracle.decision_log(True)
        try:
            res = sa.SHARP(X, ensize_K=K, rN_seed=7, logflag=False, prep=False, **kw_gpu)
            ref = oracle.SHARP(X, K=K, rN_seed=7, **kw_or)
            got, want = sa.last_decisions(), oracle.last_decisions()
        finally:
            sa.decision_log(False); oracle.decision_log(False)
        assert np.array_equal(res["pred_clusters"], ref["pred_clusters"])
        large = bool(kw_gpu)
        T = 4 if large else 1                                     # 700 cells in folds of 200: 200 / 200 / 150 / 150
        assert len(got) == K * T + T + (1 if large else 0)
        assert set(got[:, F["level"]].astype(int)) == ({0, 1, 2} if large else {0, 1})
        m = compare_logs(sa, got, want, "SHARP_%s nm=%d" % ("large" if large else "small", nm))
        if nm == 25:
            assert m["base"]["by_CH"] + m["base"]["by_height"] > 0
    assert sa.last_decisions().shape[0] == 0                      # off: cleared, and nothing more is logged
    sa.SHARP(X, ensize_K=2, rN_seed=7, logflag=False, prep=False)
    assert sa.last_decisions().shape[0] == 0


def test_unlimited_log_names_blocks_and_the_cross_block_merge(sa, oracle):
    """SHARP_unlimited over ragged blocks, one of them on the small path: block indices in the log, the level-3 row of the cross-block
    sMetaC (R/SHARP_unlimited.R:163), the batched window (tails on helper threads, in any order) and the block-by-block form alike."""
    sizes = [(5200, 0), (300, 6000), (5400, 7000), (5100, 14000)]
    blocks = [_data(oracle, 1500, n, 5, 250, c0) for n, c0 in sizes]
    oracle.decision_log(True)
    try:
        ref = oracle.SHARP_unlimited(blocks, K=3, rN_seed=2103, nthreads=8)
        want = oracle.last_decisions()
    finally:
        oracle.decision_log(False)
    import torch
    from sharp_amd import device as dev

    dblocks = [torch.from_numpy(np.ascontiguousarray(b.T.astype(np.float32))).cuda() for b in blocks]
    torch.cuda.synchronize()
    for form in ("host list", "resident, one window", "two logical devices"):
        sa.decision_log(True)
        try:
            if form == "host list":
                pred = sa.SHARP_unlimited(blocks, ensize_K=3, rN_seed=2103, viewflag=False)["pred_clusters"]
            elif form == "resident, one window":
                pred = dev.unlimited_dev(dblocks, ensize_K=3, rN_seed=2103)[0]
            else:
                pred = dev.unlimited_multi_dev(dblocks, [0, 1, 0, 1], [0, 0], ensize_K=3, rN_seed=2103)[0]
            got = sa.last_decisions()
        finally:
            sa.decision_log(False)
        assert np.array_equal(pred, ref["pred_clusters"]), form
        assert set(got[:, F["block"]].astype(int)) == {0, 1, 2, 3} and (got[:, F["level"]] == 3).sum() == 1
        compare_logs(sa, got, want, "SHARP_unlimited (" + form + ")")


def test_direct_calls_and_the_given_N_cluster(sa, oracle):
    rng = np.random.default_rng(3)
    E = rng.standard_normal((300, 40)) + rng.standard_normal((6, 40))[rng.integers(0, 6, 300)] * 2
    sa.decision_log(True); oracle.decision_log(True)
    try:
        sa.get_opt_hclust(E)
        sa.get_opt_hclust(E, N_cluster=4)
        sa.get_opt_hclust(E, sil_thre=2.0)
        oracle.get_opt_hclust(E)
        oracle.get_opt_hclust(E, N_cluster=4)
        oracle.get_opt_hclust(E, sil_thre=2.0)
        got, want = sa.last_decisions(), oracle.last_decisions()
    finally:
        sa.decision_log(False); oracle.decision_log(False)
    assert got.shape == (3, 14) and list(got[:, F["level"]]) == [-1, -1, -1] and list(got[:, F["branch"]])[1] == 3
    compare_logs(sa, got, want, "direct calls")
