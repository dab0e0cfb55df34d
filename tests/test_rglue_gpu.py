"""r/sharp_glue.c EXECUTED: the .Call entry points of the R shim, built against tests/rmock (a stand-in for the R C-API functions the shim
uses; there is no R in the image), handed R-shaped values (numeric matrices genes x cells, lists of blocks, the slots of a dgCMatrix) and
their results -- R lists -- compared with the oracle.  This covers the SEXP unpacking / packing of every entry incl.
R_sharp_unlimited_multi."""
import numpy as np
import pytest
import scipy.sparse as sps

from _rglue import Glue

pytestmark = pytest.mark.gpu
SEED = 20261003


@pytest.fixture(scope="module")
def glue():
    g = Glue()
    g.call("R_sharp_init", g.int(0))
    return g


def _ipar(K=0, hmethod=0, flashmark=0, flag=1):
    return [K, 0, 0, 0, hmethod, 0, 0, 0, 0, 0, flashmark, flag, 0]


def test_call_sharp_small_dense_and_sparse(glue, oracle):
    m, n, K = 1500, 400, 3
    X = oracle.synth_fill(SEED, m, 0, n, 4, 200)
    ref = oracle.SHARP_small(X, K=K, rN_seed=2103)
    p = int(np.ceil(np.log2(n) / 0.04))
    r = glue.call("R_sharp_SHARP", glue.matrix(X), glue.int(*_ipar(K)), glue.real(-1, 0, 2103), glue.lgl(True))
    sp = sps.csc_matrix(X)
    r2 = glue.call("R_sharp_SHARP_csc", glue.int(sp.indptr), glue.int(sp.indices), glue.real(sp.data), glue.int(m, n), glue.int(*_ipar(K)),
                   glue.real(-1, 0, 2103), glue.lgl(True))
    for res in (r, r2):
        assert np.array_equal(glue.get(res, "pred"), ref["pred_clusters"])
        assert glue.get(res, "p")[0] == p and glue.get(res, "K")[0] == K and glue.get(res, "path")[0] == 0
        viE = glue.get(res, "viE")                                   # cells x p R matrix
        assert viE.shape == (n, p)
        np.testing.assert_allclose(viE, ref["viE"], rtol=0, atol=2e-12 * np.abs(ref["viE"]).max())
        x0 = glue.get(res, "x0")
        assert x0.shape == ref["x0"].shape
        np.testing.assert_allclose(x0, ref["x0"], atol=1e-15)
    # forview = FALSE: pred only
    r3 = glue.call("R_sharp_SHARP", glue.matrix(X), glue.int(*_ipar(K)), glue.real(-1, 0, 2103), glue.lgl(False))
    assert np.array_equal(glue.get(r3, "pred"), ref["pred_clusters"]) and glue.get(r3, "viE") is None
    glue.reset()


def test_call_sharp_large(glue, oracle):
    m, n, K = 1200, 6100, 3
    X = oracle.synth_fill(SEED + 1, m, 0, n, 5, 120)
    ref = oracle.SHARP(X, K=K, rN_seed=7, nthreads=8)
    r = glue.call("R_sharp_SHARP", glue.matrix(X), glue.int(*_ipar(K)), glue.real(-1, 0, 7), glue.lgl(True))
    assert glue.get(r, "path")[0] == 1
    assert np.array_equal(glue.get(r, "pred"), ref["pred_clusters"])
    np.testing.assert_allclose(glue.get(r, "viE"), ref["viE"], rtol=0, atol=2e-12 * np.abs(ref["viE"]).max())
    glue.reset()


@pytest.mark.parametrize("devices", [(), (0,), (0, 0)])
def test_call_unlimited_multi_dense_and_sparse(glue, oracle, devices):
    """R_sharp_unlimited_multi: list of dense matrices / list of list(p, i, x, dim); devices = integer(0) (the library's own GPU), one
    GPU, two workers on logical devices of GPU 0"""
    m, K = 1300, 3
    sizes = [5200, 700, 5100]
    blocks, c0 = [], 0
    for nb in sizes:
        blocks.append(oracle.synth_fill(SEED + 2, m, c0, nb, 5, 130))
        c0 += nb
    ref = oracle.SHARP_unlimited(blocks, K=K, rN_seed=2103, nthreads=8, want_view=True)
    n = sum(sizes)
    for make in (lambda b: glue.matrix(b), lambda b: glue.csc_block(sps.csc_matrix(b))):
        r = glue.call("R_sharp_unlimited_multi", glue.list([make(b) for b in blocks]), glue.int(K, 0, 0, 0), glue.real(2103), glue.lgl(True),
                      glue.int(*devices), glue.int(0))
        assert glue.get(r, "p")[0] == ref["p"]
        assert np.array_equal(glue.get(r, "pred"), ref["pred_clusters"])
        viE = glue.get(r, "viE")
        assert viE.shape == (n, ref["p"])
        np.testing.assert_allclose(viE, ref["viE"], rtol=0, atol=2e-12 * np.abs(ref["viE"]).max())
        glue.reset()
        # view.dim = 50: what r/sharp_hip.R passes above 1e5 cells -- E1 reduced per block on the GPU (R/SHARP_unlimited.R:216-228)
        r = glue.call("R_sharp_unlimited_multi", glue.list([make(b) for b in blocks]), glue.int(K, 0, 0, 0), glue.real(2103), glue.lgl(True),
                      glue.int(*devices), glue.int(50))
        v50 = glue.get(r, "viE")
        assert v50.shape == (n, 50) and np.array_equal(glue.get(r, "pred"), ref["pred_clusters"])
        z0 = oracle.ranM(ref["p"], 50, 50 + 2103 + K + 1)
        want = np.concatenate([oracle.project(ref["viE"][a:b].T, z0, False) for a, b in zip(np.cumsum([0] + sizes[:-1]), np.cumsum(sizes))])
        np.testing.assert_allclose(v50, want, rtol=0, atol=4e-12 * np.abs(want).max())
        glue.reset()
    # the plain list entry (no devices argument) and viewflag = FALSE
    r = glue.call("R_sharp_unlimited", glue.list([glue.matrix(b) for b in blocks]), glue.int(K, 0, 0, 0), glue.real(2103), glue.lgl(False))
    assert np.array_equal(glue.get(r, "pred"), ref["pred_clusters"]) and glue.get(r, "viE") is None
    glue.reset()


def test_library_errors_become_r_errors(glue, oracle):
    """the library's status + sharp_last_error() -> error() (the reference's stop(), R/SHARP.R:171-176: a non-integer rN.seed)"""
    X = oracle.synth_fill(SEED, 300, 0, 120, 3, 40)
    with pytest.raises(RuntimeError, match="rN.seed"):
        glue.call("R_sharp_SHARP", glue.matrix(X), glue.int(*_ipar(3)), glue.real(-1, 0, 21.25), glue.lgl(False))
    glue.reset()
