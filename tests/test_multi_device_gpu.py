"""SHARP_unlimited dealt to several device slots inside ONE process (sharp_SHARP_unlimited_multi; SURVEY.md 8e,
R/SHARP_unlimited.R:125-183).  The box has one GPU: the device list names it several times, so every "device" is a slot of its own
(context, streams, workspaces, projector handles) on GPU 0 -- the code path of an 8-GPU node, with the hardware of one."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = 20261003


@pytest.fixture(scope="module")
def sa():
    import sharp_amd

    sharp_amd.init(0)
    return sharp_amd


def _blocks(oracle, sizes, m=2500, G=6, nm=250):
    out, c0 = [], 0
    for n in sizes:
        out.append(oracle.synth_fill(SEED, m, c0, n, G, nm))
        c0 += n
    return out


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0]])
def test_unlimited_on_several_slots_equals_one_device_and_the_oracle(sa, oracle, devices):
    blocks = _blocks(oracle, [5200, 5600, 5100, 5300, 5050])          # SHARP_large blocks (>= 5000 cells), ragged
    ref = oracle.SHARP_unlimited(blocks, K=3, rN_seed=2103, nthreads=8, want_view=True)
    one = sa.SHARP_unlimited(blocks, ensize_K=3, rN_seed=2103, viewflag=True)
    multi = sa.SHARP_unlimited(blocks, ensize_K=3, rN_seed=2103, viewflag=True, devices=devices)
    assert np.array_equal(one["pred_clusters"], ref["pred_clusters"])
    assert np.array_equal(multi["pred_clusters"], one["pred_clusters"])
    assert multi["N.pred_clusters"] == one["N.pred_clusters"]
    np.testing.assert_array_equal(multi["viE"], one["viE"])            # same kernels, same order: bit for bit
    np.testing.assert_allclose(multi["viE"], ref["viE"], rtol=0, atol=2e-12 * np.abs(ref["viE"]).max())


def test_unlimited_multi_through_the_dotC_convention_and_the_environment(sa, oracle, monkeypatch):
    blocks = _blocks(oracle, [900, 1100, 1000], m=1500, G=4, nm=200)   # SHARP_small blocks
    one = sa.SHARP_unlimited(blocks, ensize_K=3, rN_seed=2103, viewflag=False)
    lib = sa.lib()
    ncb = np.array([b.shape[1] for b in blocks], np.float64)
    n, m = int(ncb.sum()), blocks[0].shape[0]
    xcat = np.concatenate([np.asfortranarray(b, dtype=np.float64).ravel(order="F") for b in blocks])
    pred = np.zeros(n, np.int32)
    info = np.zeros(2, np.int32)
    dv = np.array([0, 0], np.int32)
    i = lambda v: C.byref(C.c_int(v))          # noqa: E731  (.C() passes every scalar as a length-one vector)
    status = C.c_int(-1)
    dummy = np.zeros(1)
    lib.sharp_C_SHARP_unlimited_multi(xcat.ctypes.data_as(C.POINTER(C.c_double)), i(len(blocks)), ncb.ctypes.data_as(C.POINTER(C.c_double)),
                                      i(m), i(3), i(0), i(0), i(0), C.byref(C.c_double(2103)), dv.ctypes.data_as(C.POINTER(C.c_int)), i(2),
                                      pred.ctypes.data_as(C.POINTER(C.c_int)), dummy.ctypes.data_as(C.POINTER(C.c_double)),
                                      info.ctypes.data_as(C.POINTER(C.c_int)), i(0), C.byref(status))
    assert status.value == 0, lib.sharp_last_error()
    assert np.array_equal(pred, one["pred_clusters"]) and info[0] == one["N.pred_clusters"]
    # SHARP_DEVICES in the environment sends the plain entry point down the same path
    monkeypatch.setenv("SHARP_DEVICES", "0,0")
    env = sa.SHARP_unlimited(blocks, ensize_K=3, rN_seed=2103, viewflag=False)
    assert np.array_equal(env["pred_clusters"], one["pred_clusters"])


def test_unlimited_multi_rejects_what_it_cannot_do(sa, oracle):
    blocks = _blocks(oracle, [400, 500], m=800, G=3, nm=100)
    with pytest.raises(sa.SharpError, match="needs a seed"):
        sa.SHARP_unlimited(blocks, ensize_K=3, viewflag=False, devices=[0, 0])
    with pytest.raises(sa.SharpError, match="device"):
        sa.SHARP_unlimited(blocks, ensize_K=3, rN_seed=2103, viewflag=False, devices=[0, 97])
