"""LAB build only (make -C sharp_amd/csrc LAB=1): the agglomeration forms that were measured and not adopted -- on the upper triangle of the
matrix (tools/lab/hclust_tri.inc, SHARP_HC_TRI=1) and with append-only first rounds (tools/lab/hclust_front.inc, SHARP_HC_FRONT=c) -- against
the product's kernel and the oracle.  Run on a GPU box: python -m pytest tools/lab/tests -q"""
import numpy as np
import pytest

SEED, RN = 20261003, 2103


@pytest.fixture(scope="module")
def env():
    import torch

    import sharp_amd
    from sharp_amd import device

    sharp_amd.init(0)
    return sharp_amd, device, torch


def _hc_counts(dev):
    tab = dev.profile_table()
    return tab.get("host:hclust_tasks_bulk_synchronous", (0, 0))[1], tab.get("host:hclust_tasks_sequential", (0, 0))[1]


@pytest.fixture(scope="module")
def sa():
    import sharp_amd

    sharp_amd.init(0)
    return sharp_amd


@pytest.mark.parametrize("n", [130, 700, 990])
def test_upper_triangle_and_full_matrix_agglomeration_agree(sa, oracle, n, monkeypatch):
    """hclust_tri_kernel (SHARP_HC_TRI=1: one launch per task on the upper triangle of the distance matrix, half the bytes per round) against
    the default hclust_rnn_kernel on the full matrix: same pairs, same ranks, same Lance-Williams arithmetic --
    every cutree level identical, heights to rounding, and both equal to the oracle.  Duplicated observations (exact ties) send either
    to the sequential kernel."""
    from sharp_amd import device as dev

    rng = np.random.default_rng(100 + n)
    E = rng.standard_normal((n, 40)) + np.repeat(rng.standard_normal((10, 40)) * 2.5, n // 10, axis=0)
    for hm in ["ward.D", "ward.D2", "average", "complete"]:
        monkeypatch.setenv("SHARP_HC_TRI", "1")
        dev.profile(True)
        a = sa.get_opt_hclust(E, hmethod=hm)
        assert _hc_counts(dev) == (1, 0), hm
        monkeypatch.delenv("SHARP_HC_TRI")
        dev.profile(True)
        b = sa.get_opt_hclust(E, hmethod=hm)
        assert _hc_counts(dev) == (1, 0), hm
        assert np.array_equal(a["v"], b["v"]) and np.array_equal(a["f"], b["f"]), hm
        np.testing.assert_allclose(a["height"], b["height"], rtol=1e-12, atol=1e-14)
        ref = oracle.get_opt_hclust(E, hmethod=hm)
        assert np.array_equal(a["f"], ref["f"]) and a["optN_cluster"] == ref["optN"]
        np.testing.assert_allclose(a["height"], ref["height"], rtol=1e-9, atol=1e-12)
    T = np.vstack([E[: n // 2], E[: n // 8]]).copy()            # exact duplicates -> exact ties
    monkeypatch.setenv("SHARP_HC_TRI", "1")
    dev.profile(True)
    a = sa.get_opt_hclust(T)
    assert _hc_counts(dev) == (0, 1)
    assert np.array_equal(a["v"], oracle.get_opt_hclust(T)["v"])
    dev.profile(False)


@pytest.mark.parametrize("n", [130, 700, 990, 2000])
def test_lazy_front_and_full_rewrite_agglomeration_agree(sa, oracle, n, monkeypatch):
    """hclust_front_kernel (SHARP_HC_FRONT=c: the first c rounds append new rows and the survivors' tails beside the pristine matrix instead of
    rewriting it, then one compaction and hclust_rnn_kernel's MODE 3 for the rest) against the default full rewrite: same pairs, same ranks,
    the same Lance-Williams arithmetic per entry -- every cutree level identical, heights to rounding, both equal to the oracle; exact ties
    send either to the sequential kernel.  (n = 130: no lazy round at all, the front only compacts; 2000: a base-clustering task.)"""
    from sharp_amd import device as dev

    rng = np.random.default_rng(300 + n)
    E = rng.standard_normal((n, 40)) + np.repeat(rng.standard_normal((10, 40)) * 2.5, n // 10, axis=0)
    monkeypatch.setenv("SHARP_HC_SPLIT", "0")                     # (one launch per task also at n >= 1000, where few tasks would go round by round)
    for hm in ["ward.D", "ward.D2", "average", "complete"]:
        for c in (4, 2, 9):
            monkeypatch.setenv("SHARP_HC_FRONT", str(c))
            dev.profile(True)
            a = sa.get_opt_hclust(E, hmethod=hm)
            assert _hc_counts(dev) == (1, 0), (hm, c)
            monkeypatch.delenv("SHARP_HC_FRONT")
            dev.profile(True)
            b = sa.get_opt_hclust(E, hmethod=hm)
            assert _hc_counts(dev) == (1, 0), hm
            assert np.array_equal(a["v"], b["v"]) and np.array_equal(a["f"], b["f"]), (hm, c)
            np.testing.assert_allclose(a["height"], b["height"], rtol=1e-12, atol=1e-14)
            if hm != "ward.D":
                break                                             # (the other linkages: one setting)
        ref = oracle.get_opt_hclust(E, hmethod=hm)
        assert np.array_equal(a["f"], ref["f"]) and a["optN_cluster"] == ref["optN"]
        np.testing.assert_allclose(a["height"], ref["height"], rtol=1e-9, atol=1e-12)
    T = np.vstack([E[: n // 2], E[: n // 8]]).copy()            # exact duplicates -> exact ties
    monkeypatch.setenv("SHARP_HC_FRONT", "4")
    dev.profile(True)
    a = sa.get_opt_hclust(T)
    assert _hc_counts(dev) == (0, 1)
    assert np.array_equal(a["v"], oracle.get_opt_hclust(T)["v"])
    dev.profile(False)


def test_upper_triangle_agglomeration_at_cfg2_size(env, monkeypatch):
    """BASELINE.json configs[1] (50 000 x 20 000, ensize.K = 15: 375 base tasks of 2000 cells in two chunks of one task per CU) with the
    upper-triangle agglomeration kernel (SHARP_HC_TRI=1) and with the default full-matrix one: every task done by the bulk-synchronous
    kernel either way, identical labels."""
    sa, dev, torch = env
    n, m, K = 50000, 20000, 15
    dX = torch.empty((n, m), dtype=torch.float32, device="cuda")
    dev.synth_fill(dX, SEED, 0)
    dev.profile(True)
    pred, info = dev.SHARP_dev(dX, ensize_K=K, rN_seed=RN)
    tab = dev.profile_table()
    assert tab.get("host:hclust_tasks_sequential", (0, 0))[1] <= 25 + 1        # (the 25 wMetaC similarity tasks and the sMetaC one have exact ties)
    monkeypatch.setenv("SHARP_HC_TRI", "1")
    dev.profile(True)
    pred_tri, info_tri = dev.SHARP_dev(dX, ensize_K=K, rN_seed=RN)
    tab_tri = dev.profile_table()
    dev.profile(False)
    assert tab_tri.get("host:hclust_tasks_bulk_synchronous", (0, 0))[1] >= 375
    assert np.array_equal(pred, pred_tri) and info["N.pred_cluster"] == info_tri["N.pred_cluster"]
