"""SHARP_unlimited3 (R/SHARP_unlimited3.R:29-235): a DIRECTORY of partitions streamed disk -> pinned memory -> HBM through a
ring of buffers ahead of the clustering -- dense block files and the packed sparse format (counts: 4 bytes per non-zero, expanded on the
device) -- against the oracle's SHARP_unlimited on the same partitions, label for label."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = 20261003
_ORACLE_CACHE = {}


@pytest.fixture(scope="module")
def sa():
    import sharp_amd

    sharp_amd.init(0)
    return sharp_amd


@pytest.fixture
def digit_free_dir():
    """The reference orders partitions by the first number in the FULL path (R/SHARP_unlimited3.R:60), so a digit in a
    parent directory (pytest's tmp_path has one) makes every key equal; use a path without digits."""
    import shutil
    import string
    import tempfile

    rng = np.random.default_rng()
    name = "sharpblk_" + "".join(rng.choice(list(string.ascii_lowercase), 12))
    d = os.path.join(tempfile.gettempdir(), name)
    if any(ch.isdigit() for ch in d):
        pytest.skip("temporary directory path contains digits")
    os.mkdir(d)
    yield d
    shutil.rmtree(d, ignore_errors=True)


@pytest.mark.parametrize("fmt", ["dense", "packed"])
def test_sharp_unlimited3_streams_a_directory_of_blocks(sa, oracle, digit_free_dir, fmt):
    """Same result as SHARP_unlimited on the same partitions and as the oracle, read from block files in the order of the number in their
    name (part_10 after part_2); with the view outputs (block after block: every block's E1 rows) and without (the blocks that have arrived
    taken together as one pipelined batch)."""
    from sharp_amd import blocks as B

    m, G, nm = 1500, 6, 150
    sizes = [5300, 5050, 5200, 5100]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    parts = [oracle.synth_fill(SEED, m, int(offs[i]), sizes[i], G, nm) for i in range(len(sizes))]
    d = digit_free_dir
    names = ["part_1.blk", "part_2.blk", "part_10.blk", "part_100.blk"]      # alphabetical order would be 1, 10, 100, 2
    for name, X in zip(names, parts):
        B.write_block(os.path.join(d, name), X, fmt)
    assert [os.path.basename(f) for f in B.list_block_files(d + "/")] == names
    h = B.read_header(os.path.join(d, "part_10.blk"))
    assert (h["genes"], h["cells"], h["ld"], h["version"]) == (m, 5200, m, 1 if fmt == "dense" else 2)
    nd = {"dir": d, "ncells": int(offs[-1]), "ngenes": m}
    if "ref" not in _ORACLE_CACHE:
        _ORACLE_CACHE["ref"] = oracle.SHARP_unlimited(parts, K=3, rN_seed=2103, nthreads=8, want_view=True)
    ref = _ORACLE_CACHE["ref"]
    res3 = sa.SHARP_unlimited3(nd, ensize_K=3, rN_seed=2103)
    assert np.array_equal(res3["pred_clusters"], ref["pred_clusters"])
    np.testing.assert_allclose(res3["viE"], ref["viE"], rtol=0, atol=2e-12 * np.abs(ref["viE"]).max())
    payload = sum(B.read_header(os.path.join(d, f))["payload"] for f in names)
    assert res3["bytes_streamed"] == payload
    if fmt == "dense":
        assert payload == sum(sizes) * m * 4
    else:
        nnz = sum(int(np.count_nonzero(X)) for X in parts)
        assert payload <= 4 * nnz + (sum(sizes) + len(sizes)) * 8 + 64 * len(sizes)          # 16-bit values and row indices, the column pointers
    res = sa.SHARP_unlimited(parts, ensize_K=3, rN_seed=2103)
    assert np.array_equal(res3["pred_clusters"], res["pred_clusters"]) and np.array_equal(res3["viE"], res["viE"])
    grouped = sa.SHARP_unlimited3(nd, ensize_K=3, rN_seed=2103, viewflag=False)                           # the arrived blocks together
    assert np.array_equal(grouped["pred_clusters"], ref["pred_clusters"]) and grouped["bytes_streamed"] == payload
    assert grouped["read_seconds"] > 0 and grouped["wait_seconds"] >= 0
    with pytest.raises(sa.SharpError, match="should be a folder"):
        sa.SHARP_unlimited3({"dir": os.path.join(d, "missing"), "ncells": 10, "ngenes": m})


def test_packed_files_of_doubles_and_a_small_block(sa, oracle, digit_free_dir):
    """TPM-like values (not fp32-exact): 64-bit values in the file, an fp64 block on the device, the reference's numbers; one partition below
    5 000 cells (the small path) between two large ones; the streamer's ring of two."""
    from sharp_amd import blocks as B

    m = 1500
    sizes = [5200, 700, 5300]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    parts = []
    for i, nb in enumerate(sizes):
        X = oracle.synth_fill(77, m, int(offs[i]), nb, 5, 150)
        parts.append(X / np.maximum(X.sum(0, keepdims=True), 1.0) * 1e6)
    d = digit_free_dir
    for i, X in enumerate(parts):
        B.write_block(os.path.join(d, "p%d.blk" % (i + 1)), X)
        assert B.read_header(os.path.join(d, "p%d.blk" % (i + 1)))["f64"]
    ref = oracle.SHARP_unlimited(parts, K=3, rN_seed=7, nthreads=8)
    res = sa.SHARP_unlimited3({"dir": d, "ncells": int(offs[-1]), "ngenes": m}, ensize_K=3, rN_seed=7, viewflag=False)
    assert np.array_equal(res["pred_clusters"], ref["pred_clusters"])
    st = B.BlockStreamer(B.list_block_files(d), ring=2)
    got = [(i, tuple(x.shape), str(x.dtype)) for i, h, x in st]
    assert got == [(0, (5200, m), "torch.float64"), (1, (700, m), "torch.float64"), (2, (5300, m), "torch.float64")]
