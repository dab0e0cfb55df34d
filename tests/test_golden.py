"""Golden fixtures (tests/golden/oracle_small.npz, made by tests/golden/make_golden.py): the oracle must keep
reproducing them (CPU), and the HIP path must reproduce them through the C ABI (GPU)."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SEED = 20261003


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "oracle_small.npz"))


def test_oracle_reproduces_known_r_outputs(oracle, gold):
    np.testing.assert_allclose(oracle.runif(1, 3), gold["R_runif_seed1"], atol=5e-8)
    np.testing.assert_allclose(oracle.runif(42, 2), gold["R_runif_seed42"], atol=5e-8)
    assert oracle.sample_perm(42, 10).tolist() == gold["R_sample10_seed42"].tolist()
    assert oracle.sample_perm(123, 10).tolist() == gold["R_sample10_seed123"].tolist()


def test_oracle_reproduces_golden_vectors(oracle, gold):
    m, n = gold["synth_X"].shape
    X = oracle.synth_fill(SEED, m, 0, n, 3, 150)
    assert np.array_equal(X.astype(np.float32), gold["synth_X"])
    p = int(np.ceil(np.log2(n) / 0.04))
    t = oracle.ranM(m, p, 2154)
    gi, ci = np.nonzero(t)
    assert np.array_equal(gi, gold["ranM_gene"]) and np.array_equal(ci, gold["ranM_col"])
    assert np.array_equal(t[gi, ci], gold["ranM_sign"])
    E = oracle.project(X, t, True)
    assert np.array_equal(E, gold["E"])
    r = oracle.get_opt_hclust(E)
    assert np.array_equal(r["f"], gold["hc_f"]) and np.array_equal(r["v"], gold["hc_v"])
    np.testing.assert_allclose(r["msil"], gold["hc_msil"], rtol=0, atol=1e-15)
    s = oracle.SHARP_small(X, K=3, rN_seed=2103)
    assert np.array_equal(s["pred_clusters"], gold["small_pred"]) and np.array_equal(s["enrp"], gold["small_enrp"])
    big = oracle.synth_fill(SEED, m, 0, 260, 3, 150)
    L = oracle.SHARP(big, K=3, base_ncells=100, partition_ncells=80, rN_seed=2103)
    assert np.array_equal(L["pred_clusters"], gold["large_pred"])


@pytest.mark.gpu
def test_hip_path_reproduces_golden_vectors(gold):
    import sharp_amd

    sharp_amd.init(0)
    X = gold["synth_X"].astype(np.float64)
    m, n = X.shape
    p = int(np.ceil(np.log2(n) / 0.04))
    pr = sharp_amd.ranM2(m, p, 2154)
    g, c, s = pr.triplets(0)
    assert np.array_equal(g, gold["ranM_gene"]) and np.array_equal(c, gold["ranM_col"]) and np.array_equal(s, gold["ranM_sign"])
    E = pr.project(X, True)
    np.testing.assert_allclose(E, gold["E"], rtol=0, atol=2e-12 * np.abs(gold["E"]).max())
    r = sharp_amd.get_opt_hclust(gold["E"])
    assert np.array_equal(r["f"], gold["hc_f"]) and np.array_equal(r["v"], gold["hc_v"])
    np.testing.assert_allclose(r["height"], gold["hc_height"], rtol=1e-9)
    np.testing.assert_allclose(r["msil"], gold["hc_msil"], atol=1e-10)
    np.testing.assert_allclose(r["CHind"], gold["hc_CH"], rtol=1e-8)
    w = sharp_amd.wMetaC(gold["small_enrp"], sil_thre=0.35, debug=True)
    np.testing.assert_allclose(w["w1"], gold["wm_w1"], rtol=1e-13)
    np.testing.assert_allclose(w["S"], gold["wm_S"], rtol=1e-12, atol=1e-15)
    assert np.array_equal(w["finalC"], gold["wm_finalC"])
    res = sharp_amd.SHARP(X, ensize_K=3, rN_seed=2103, logflag=False, prep=False)
    assert np.array_equal(res["pred_clusters"], gold["small_pred"])
    np.testing.assert_allclose(res["x0"], gold["small_x0"], atol=1e-15)
    from oracle import pyoracle as orc

    big = orc.synth_fill(SEED, m, 0, 260, 3, 150)
    L = sharp_amd.SHARP(big, ensize_K=3, base_ncells=100, partition_ncells=80, rN_seed=2103, logflag=False, prep=False)
    assert np.array_equal(L["pred_clusters"], gold["large_pred"])
