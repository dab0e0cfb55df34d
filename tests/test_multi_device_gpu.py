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


def test_unlimited_on_several_slots_equals_one_device_and_the_oracle(sa, oracle):
    blocks = _blocks(oracle, [5200, 5600, 5100, 5300, 5050])          # SHARP_large blocks (>= 5000 cells), ragged
    ref = oracle.SHARP_unlimited(blocks, K=3, rN_seed=2103, nthreads=8, want_view=True)
    one = sa.SHARP_unlimited(blocks, ensize_K=3, rN_seed=2103, viewflag=True)
    assert np.array_equal(one["pred_clusters"], ref["pred_clusters"])
    np.testing.assert_allclose(one["viE"], ref["viE"], rtol=0, atol=2e-12 * np.abs(ref["viE"]).max())
    for devices in ([0, 0], [0, 0, 0]):                               # two and three logical devices on the one GPU
        multi = sa.SHARP_unlimited(blocks, ensize_K=3, rN_seed=2103, viewflag=True, devices=devices)
        assert np.array_equal(multi["pred_clusters"], one["pred_clusters"])
        assert multi["N.pred_clusters"] == one["N.pred_clusters"]
        np.testing.assert_array_equal(multi["viE"], one["viE"])        # same kernels, same order: bit for bit


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


def test_three_slots_seven_ragged_blocks_host_sparse_and_resident_inputs(sa, oracle):
    """three logical devices x seven ragged blocks; the same blocks as dense host matrices, as a list of sparse
    (dgCMatrix-like) blocks and as blocks already resident on their GPU give identical labels and bit-identical viE -- and the labels of
    oracle.SHARP_unlimited.  The upload of block b + W runs under the clustering of block b (sharp_multi_timeline)."""
    import scipy.sparse as sp
    import torch
    from sharp_amd import device as dev

    sizes = [5200, 5600, 5100, 5300, 5050, 5400, 5150]
    blocks = _blocks(oracle, sizes)
    ref = oracle.SHARP_unlimited(blocks, K=3, rN_seed=2103, nthreads=8, want_view=True)
    one = sa.SHARP_unlimited(blocks, ensize_K=3, rN_seed=2103, viewflag=True)
    assert np.array_equal(one["pred_clusters"], ref["pred_clusters"])
    devices = [0, 0, 0]
    multi = sa.SHARP_unlimited(blocks, ensize_K=3, rN_seed=2103, viewflag=True, devices=devices)
    tl = dev.multi_timeline()
    assert tl.shape[0] == len(sizes) and sorted(tl[:, 1].astype(int).tolist()) == list(range(len(sizes)))
    assert np.all(tl[:, 3] >= tl[:, 2]) and np.all(tl[:, 5] >= tl[:, 4]) and np.all(tl[:, 4] >= tl[:, 3])   # clustered after its upload
    assert set(tl[:, 0].astype(int).tolist()) == {0, 1, 2}
    # a worker's second block was uploaded before its first one had been clustered: the upload is hidden
    for w in range(3):
        rows = tl[tl[:, 0] == w]
        rows = rows[np.argsort(rows[:, 1])]
        assert rows[1, 2] < rows[0, 5]
    sparse = sa.SHARP_unlimited([sp.csc_matrix(b) for b in blocks], ensize_K=3, rN_seed=2103, viewflag=True, devices=devices)
    sparse_one = sa.SHARP_unlimited([sp.csr_matrix(b) for b in blocks], ensize_K=3, rN_seed=2103, viewflag=True)   # one device, pipelined uploads
    dblocks = [torch.from_numpy(np.ascontiguousarray(b.T.astype(np.float32))).cuda() for b in blocks]
    torch.cuda.synchronize()
    pred, npred, p, viE = dev.unlimited_multi_dev(dblocks, [b % 3 for b in range(len(sizes))], devices, ensize_K=3, rN_seed=2103, viewflag=True)
    for got in (multi, sparse, sparse_one):
        assert np.array_equal(got["pred_clusters"], one["pred_clusters"]) and got["N.pred_clusters"] == one["N.pred_clusters"]
        np.testing.assert_array_equal(got["viE"], one["viE"])
    assert np.array_equal(pred, one["pred_clusters"]) and npred == one["N.pred_clusters"] and p == one["viE"].shape[1]
    np.testing.assert_array_equal(viE, one["viE"])
    # uneven ownership: every resident block on the LAST listed device
    pred2, npred2, _, _ = dev.unlimited_multi_dev(dblocks, [2] * len(sizes), devices, ensize_K=3, rN_seed=2103)
    assert np.array_equal(pred2, one["pred_clusters"])


def test_sparse_tpm_blocks_and_the_dotC_form(sa, oracle):
    """Ragged sparse blocks of TPM-like (non-fp32-exact) values: stored as fp64 on the GPU like the dense entry stores them; labels as
    the dense call and the oracle; the .C() form of the sparse list (sharp_C_SHARP_unlimited_csc)."""
    import scipy.sparse as sp

    blocks = _blocks(oracle, [900, 1300, 1000], m=1500, G=4, nm=200)
    tpm = [b / np.maximum(b.sum(0, keepdims=True), 1.0) * 1e6 for b in blocks]
    ref = oracle.SHARP_unlimited(tpm, K=3, rN_seed=2103, nthreads=8, want_view=False)
    dense = sa.SHARP_unlimited(tpm, ensize_K=3, rN_seed=2103, viewflag=True)
    sparse = sa.SHARP_unlimited([sp.csc_matrix(b) for b in tpm], ensize_K=3, rN_seed=2103, viewflag=True, devices=[0, 0])
    assert sa.lib().sharp_x_storage() == 64
    assert np.array_equal(dense["pred_clusters"], ref["pred_clusters"])
    assert np.array_equal(sparse["pred_clusters"], dense["pred_clusters"])
    np.testing.assert_array_equal(sparse["viE"], dense["viE"])
    # the .C() convention: every argument a pointer, the blocks' slots concatenated
    lib = sa.lib()
    cs = [sp.csc_matrix(b) for b in blocks]
    pcat = np.concatenate([c.indptr.astype(np.int32) for c in cs])
    icat = np.concatenate([c.indices.astype(np.int32) for c in cs])
    xcat = np.concatenate([c.data.astype(np.float64) for c in cs])
    ncb = np.array([b.shape[1] for b in blocks], np.float64)
    n, m = int(ncb.sum()), blocks[0].shape[0]
    pred = np.zeros(n, np.int32)
    info = np.zeros(2, np.int32)
    dv = np.array([0], np.int32)
    i = lambda v: C.byref(C.c_int(v))          # noqa: E731
    status = C.c_int(-1)
    dummy = np.zeros(1)
    lib.sharp_C_SHARP_unlimited_csc(pcat.ctypes.data_as(C.POINTER(C.c_int)), icat.ctypes.data_as(C.POINTER(C.c_int)),
                                    xcat.ctypes.data_as(C.POINTER(C.c_double)), i(len(blocks)), ncb.ctypes.data_as(C.POINTER(C.c_double)),
                                    i(m), i(3), i(0), i(0), i(0), C.byref(C.c_double(2103)), dv.ctypes.data_as(C.POINTER(C.c_int)), i(0),
                                    pred.ctypes.data_as(C.POINTER(C.c_int)), dummy.ctypes.data_as(C.POINTER(C.c_double)),
                                    info.ctypes.data_as(C.POINTER(C.c_int)), i(0), C.byref(status))
    assert status.value == 0, lib.sharp_last_error()
    counts = sa.SHARP_unlimited(blocks, ensize_K=3, rN_seed=2103, viewflag=False)
    assert np.array_equal(pred, counts["pred_clusters"]) and info[0] == counts["N.pred_clusters"]


def test_device_lists_may_change_between_calls_and_trim_reaches_every_slot(sa, oracle, monkeypatch):
    """a worker's context is found by (device, occurrence), so {0, 0} followed by {0} or {0, 0, 0} works; sharp_trim gives
    back what EVERY slot holds; an unseeded call with SHARP_DEVICES in the environment stays on the caller's GPU instead of failing."""
    import torch

    blocks = _blocks(oracle, [700, 800, 900, 650], m=1200, G=3, nm=150)
    one = sa.SHARP_unlimited(blocks, ensize_K=3, rN_seed=7, viewflag=False)
    for devices in ([0, 0], [0], [0, 0, 0], [0, 0]):
        got = sa.SHARP_unlimited(blocks, ensize_K=3, rN_seed=7, viewflag=False, devices=devices)
        assert np.array_equal(got["pred_clusters"], one["pred_clusters"])
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    assert sa.lib().sharp_trim() == 0
    free1 = torch.cuda.mem_get_info()[0]
    assert free1 >= free0                                     # (the worker slots' resident copies and workspaces came back)
    monkeypatch.setenv("SHARP_DEVICES", "0,0")
    unseeded = sa.SHARP_unlimited(blocks, ensize_K=3, viewflag=False)          # rN.seed = NULL: the reference's default call
    assert unseeded["pred_clusters"].shape == one["pred_clusters"].shape


def test_shutdown_and_reinit_reach_every_slot(sa, oracle):
    """sharp_shutdown destroys the streams of the caller's slot AND of every worker / helper slot earlier calls have left behind; after
    sharp_init the same calls run again (streams are recreated on first use of a slot, workspaces are kept) and give the same labels."""
    blocks = _blocks(oracle, [700, 800, 900], m=1200, G=3, nm=150)
    one = sa.SHARP_unlimited(blocks, ensize_K=3, rN_seed=7, viewflag=False)
    two = sa.SHARP_unlimited(blocks, ensize_K=3, rN_seed=7, viewflag=False, devices=[0, 0])
    assert np.array_equal(one["pred_clusters"], two["pred_clusters"])
    lib = sa.lib()
    for _ in range(2):
        sa.shutdown()
        assert lib.sharp_synchronize() != 0                   # no context: every entry fails loudly until sharp_init
        sa.init(0)
        again = sa.SHARP_unlimited(blocks, ensize_K=3, rN_seed=7, viewflag=False, devices=[0, 0])
        assert np.array_equal(again["pred_clusters"], one["pred_clusters"])
        alone = sa.SHARP(blocks[0], ensize_K=3, rN_seed=7, logflag=False)
        assert alone["pred_clusters"].shape == (700,)
