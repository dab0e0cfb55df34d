"""Full BASELINE.json sizes, checked through size-independent properties (the oracle would need minutes there):
determinism, first-appearance / size-ordered numbering, recovery of the planted clusters, invariance of the
final partition to the order of the blocks."""
import ctypes as C

import numpy as np
import pytest
from sklearn.metrics import adjusted_rand_score

pytestmark = pytest.mark.gpu
SEED, RN = 20261003, 2103


@pytest.fixture(scope="module")
def env():
    import torch

    import sharp_amd
    from sharp_amd import device

    sharp_amd.init(0)
    return sharp_amd, device, torch


def test_cfg2_sharp_large_properties(env):
    sa, dev, torch = env
    n, m = 50000, 20000                       # BASELINE.json configs[1]
    dX = torch.empty((n, m), dtype=torch.float32, device="cuda")
    dev.synth_fill(dX, SEED, 0)
    truth = dev.synth_labels(SEED, 0, n)
    p1, info = dev.SHARP_dev(dX, ensize_K=15, rN_seed=RN)
    p2, _ = dev.SHARP_dev(dX, ensize_K=15, rN_seed=RN)
    assert info["path"] == "SHARP_large" and info["reduced.dim"] == 391 and info["ensize.K"] == 15
    assert np.array_equal(p1, p2)                                         # bit-reproducible end to end
    first = [int(np.argmax(p1 == j)) for j in range(1, p1.max() + 1)]
    assert first == sorted(first) and p1.min() == 1 and info["N.pred_cluster"] == p1.max()
    assert adjusted_rand_score(truth, p1) > 0.95
    # < 10-cell clusters are merged for n > 1e4 (R/SHARP.R:816-825): at most one such cluster can remain
    assert (np.bincount(p1)[1:] < 10).sum() <= 1
    # a different seed gives a different ensemble but the same biology
    p3, _ = dev.SHARP_dev(dX, ensize_K=15, rN_seed=RN + 1)
    assert adjusted_rand_score(p1, p3) > 0.9


def test_cfg3_shape_unlimited_block_order_invariance(env):
    sa, dev, torch = env
    lib = sa.lib()
    nb, m = 50000, 20000                      # two of cfg3's ten 50 000-cell blocks, K = 5
    blocks = []
    for b in range(2):
        x = torch.empty((nb, m), dtype=torch.float32, device="cuda")
        dev.synth_fill(x, SEED, b * nb)
        blocks.append(x)

    def run(order):
        ptrs = (C.c_void_p * 2)(*[blocks[b].data_ptr() for b in order])
        ncb = np.array([nb, nb], np.int64)
        ldb = np.array([m, m], np.int64)
        pred = np.zeros(2 * nb, np.int32)
        npred, pu = C.c_int(), C.c_int()
        torch.cuda.synchronize()
        rc = lib.sharp_SHARP_unlimited_dev(ptrs, ncb.ctypes.data_as(C.POINTER(C.c_longlong)),
                                           ldb.ctypes.data_as(C.POINTER(C.c_longlong)), 2, m, 5, 0, 0, 0, C.c_double(RN),
                                           pred.ctypes.data_as(C.POINTER(C.c_int)), C.byref(npred), C.byref(pu))
        assert rc in (0, 16, 32, 48), lib.sharp_last_error()
        return pred, npred.value, pu.value

    pa, na, p = run([0, 1])
    pb, nb_, _ = run([1, 0])
    assert p == int(np.ceil(np.log2(2 * nb) / 0.04))
    sizes = np.bincount(pa)[1:]
    assert np.all(np.diff(sizes) <= 0) and na == pa.max()                 # ids by decreasing size
    pb_reordered = np.concatenate([pb[nb:], pb[:nb]])
    assert adjusted_rand_score(pa, pb_reordered) >= 0.99                  # same partition whatever the block order
    truth = np.concatenate([dev.synth_labels(SEED, 0, nb), dev.synth_labels(SEED, nb, nb)])
    assert adjusted_rand_score(truth, pa) > 0.9
