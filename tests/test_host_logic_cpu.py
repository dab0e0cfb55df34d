"""Host-side pieces of the Python mirror that need no GPU: Holm adjustment, one-hot x0, reference seed rules."""
import numpy as np

from sharp_amd import api
from sharp_amd import dist as sdist


def test_p_adjust_holm_matches_r():
    # p.adjust(c(0.01, 0.04, 0.03, 0.005), "holm") -> 0.03 0.06 0.06 0.02 ; ties and the cap at 1
    assert np.allclose(api._p_adjust_holm([0.01, 0.04, 0.03, 0.005]), [0.03, 0.06, 0.06, 0.02])
    assert np.allclose(api._p_adjust_holm([0.5, 0.5, 0.9]), [1.0, 1.0, 1.0])
    assert api._p_adjust_holm([]).size == 0


def test_one_hot_x0():
    pred = np.array([2, 1, 2, 3], np.int32)
    x0 = api._one_hot(pred, 3)
    x0 = x0.toarray() if hasattr(x0, "toarray") else x0
    assert x0.shape == (4, 3) and np.array_equal(x0.argmax(1) + 1, pred) and x0.sum() == 4


def test_reduced_dimension_and_block_owner_rules():
    # p = ceiling(log2(ncells) / 0.2^2) from the TOTAL number of cells (R/SHARP_unlimited.R:65-66)
    assert sdist.global_reduced_dim(50000) == 391 and sdist.global_reduced_dim(500000) == 474
    assert sdist.global_reduced_dim(1306127) == 508 and sdist.global_reduced_dim(10000000) == 582
    assert [sdist.block_owner(b, 4) for b in range(6)] == [0, 1, 2, 3, 0, 1]
