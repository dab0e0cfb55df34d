"""r/sharp_glue.c -- the .Call shim BASELINE.json's north_star names -- compiled (gcc -Wall -Werror) against tests/rmock, a stand-in for the
R C-API functions it uses (no R in the image), and run as far as a GPU-less box allows: registration table, argument validation, the
error path (R's error() = longjmp out of the shim), PROTECT balance.  The GPU half is tests/test_rglue_gpu.py."""
import numpy as np
import pytest
import scipy.sparse as sps

from _rglue import Glue


@pytest.fixture(scope="module")
def glue():
    return Glue()


def test_registration_table(glue):
    for name, nargs in [("R_sharp_init", 1), ("R_sharp_trim", 0), ("R_sharp_SHARP", 4), ("R_sharp_SHARP_csc", 7), ("R_sharp_unlimited", 4),
                        ("R_sharp_unlimited_multi", 6)]:
        assert glue.L.rmock_registered(name.encode()) == nargs
        assert hasattr(glue.L, name)


def test_empty_list_is_the_reference_error(glue):
    """R/SHARP_unlimited.R:33-35"""
    for fn, extra in (("R_sharp_unlimited", ()), ("R_sharp_unlimited_multi", (glue.int(), glue.int(0)))):
        with pytest.raises(RuntimeError, match="No expression data is provided!"):
            glue.call(fn, glue.list([]), glue.int(5, 0, 0, 0), glue.real(2103), glue.lgl(False), *extra)
    glue.reset()


def test_blocks_must_share_the_gene_axis(glue):
    a, b = np.zeros((30, 8)), np.zeros((31, 8))
    with pytest.raises(RuntimeError, match="LIST of partitioned"):
        glue.call("R_sharp_unlimited_multi", glue.list([glue.matrix(a), glue.matrix(b)]), glue.int(5, 0, 0, 0), glue.real(2103), glue.lgl(False), glue.int(), glue.int(0))
    with pytest.raises(RuntimeError, match="LIST of partitioned"):
        glue.call("R_sharp_unlimited", glue.list([glue.matrix(a), glue.int(1, 2, 3)]), glue.int(5, 0, 0, 0), glue.real(2103), glue.lgl(False))
    glue.reset()


def test_malformed_sparse_blocks_are_refused_before_any_pointer_is_used(glue):
    """a hand-built list(p, i, x, dim) must not make the upload threads read past the R vectors"""
    sp = sps.random(40, 12, density=0.3, format="csc", random_state=1)
    good = dict(p=sp.indptr.copy(), i=sp.indices.copy(), x=sp.data.copy(), dim=np.array(sp.shape))

    def block(**over):
        d = dict(good, **over)
        return glue.list([glue.int(d["p"]), glue.int(d["i"]), glue.real(d["x"]), glue.int(d["dim"])], ["p", "i", "x", "dim"])

    def run(b):
        return glue.call("R_sharp_unlimited_multi", glue.list([b]), glue.int(5, 0, 0, 0), glue.real(2103), glue.lgl(False), glue.int(), glue.int(0))

    p_bad0 = good["p"].copy(); p_bad0[0] = 1
    p_dec = good["p"].copy(); p_dec[3] = p_dec[2] - 1 if p_dec[2] > 0 else p_dec[4] + 1
    i_oob = good["i"].copy(); i_oob[5] = 40
    for b, msg in [(block(p=p_bad0), "does not start at 0"), (block(p=p_dec), "not non-decreasing"),
                   (block(i=good["i"][:-3]), "equal length"), (block(i=good["i"][:-3], x=good["x"][:-3]), "equal length"),
                   (block(i=i_oob), "outside 0"), (block(p=good["p"][:-1]), "LIST of partitioned"),
                   (block(dim=np.array([40, 12, 1])), "LIST of partitioned")]:
        with pytest.raises(RuntimeError, match=msg):
            run(b)
    # a well-formed block gets past the checks: on a GPU-less box the library then reports the missing device as an R error
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="device"):
            run(block())
    glue.reset()


def test_flashmark_with_another_method_is_the_reference_error(glue):
    """R/get_opt_hclust.R:79: `x == "ward.D" || "ward.D2"` stops for any other method"""
    ipar = glue.int(0, 0, 0, 0, 3, 0, 0, 0, 0, 0, 1, 1, 0)              # hmethod 3, flashmark TRUE
    with pytest.raises(RuntimeError, match="invalid 'y' type"):
        glue.call("R_sharp_SHARP", glue.matrix(np.zeros((20, 10))), ipar, glue.real(-1, 0, 0.5), glue.lgl(False))
    glue.reset()
