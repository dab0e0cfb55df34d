"""Host-side logic of the SHARP_unlimited3 block files (no GPU): format round trip, validation, reference file order."""
import os

import numpy as np
import pytest

from sharp_amd import blocks as B


def test_block_file_round_trip(tmp_path):
    rng = np.random.default_rng(3)
    X = rng.poisson(0.3, size=(1003, 17)).astype(np.float64)       # genes x cells, genes not a multiple of 4
    f = str(tmp_path / "a.blk")
    B.write_block(f, X, "dense")
    h = B.read_header(f)
    assert (h["version"], h["genes"], h["cells"], h["ld"], h["f64"]) == (1, 1003, 17, 1004, False)
    raw = np.fromfile(f, np.float32, offset=B.HEADER_BYTES).reshape(17, 1004)
    assert np.array_equal(raw[:, :1003], X.T.astype(np.float32)) and np.all(raw[:, 1003:] == 0)
    assert np.array_equal(B.read_block(f), X)
    with open(f, "r+b") as fh:                                       # truncated payload
        fh.truncate(os.path.getsize(f) - 4)
    with pytest.raises(ValueError, match="truncated"):
        B.read_header(f)
    g = str(tmp_path / "b.blk")
    open(g, "wb").write(b"not a block file" * 8)
    with pytest.raises(ValueError, match="bad header"):
        B.read_header(g)


def test_packed_block_file_takes_the_narrowest_exact_types(tmp_path):
    """Version 2 (the three slots of a dgCMatrix): counts -> 8- or 16-bit values, 16-bit row indices up to 65 536 genes: 3-4 bytes per non-zero;
    other fp32-exact values -> float; anything else -> double (an fp64 block on the device); more genes -> 32-bit indices.  Round trips are exact."""
    import scipy.sparse as sp

    rng = np.random.default_rng(4)
    X = np.where(rng.random((1003, 40)) < 0.1, rng.integers(1, 400, (1003, 40)), 0).astype(np.float64)
    X[5, 3] = 65535
    f = str(tmp_path / "p.blk")
    for src in (X, sp.csr_matrix(X), sp.csc_matrix(X)):
        B.write_block(f, src)                                        # auto: 10 % non-zeros -> packed
        h = B.read_header(f)
        nnz = int(np.count_nonzero(X))
        assert (h["version"], h["genes"], h["cells"], h["nnz"], h["idx_bits"], h["val_bits"], h["f64"]) == (2, 1003, 40, nnz, 16, 16, False)
        assert os.path.getsize(f) <= B.HEADER_BYTES + 41 * 8 + 4 * nnz + 48
        assert np.array_equal(B.read_block(f), X)
    for scale, bits, f64 in ((0.5, 32, False), (1.0 / 3.0, 64, True)):
        B.write_block(f, X * scale)
        h = B.read_header(f)
        assert (h["val_bits"], h["f64"], h["ld"]) == (bits, f64, 1004)
        assert np.array_equal(B.read_block(f), X * scale)
    Xs = np.minimum(X, 255)                                           # typical UMI counts: 8-bit values, 3 bytes per non-zero
    B.write_block(f, Xs)
    h = B.read_header(f)
    assert h["val_bits"] == 8 and os.path.getsize(f) <= B.HEADER_BYTES + 41 * 8 + 3 * int(np.count_nonzero(Xs)) + 48 and np.array_equal(B.read_block(f), Xs)
    Xb = X.copy(); Xb[7, 7] = 65536                                 # one count too large for 16 bits
    B.write_block(f, Xb)
    assert B.read_header(f)["val_bits"] == 32 and np.array_equal(B.read_block(f), Xb)
    big = sp.csc_matrix((np.array([3.0, 4.0]), (np.array([70000, 2]), np.array([0, 1]))), shape=(70001, 2))
    B.write_block(f, big)
    h = B.read_header(f)
    assert h["idx_bits"] == 32 and np.array_equal(B.read_block(f), big.toarray())
    dense = rng.integers(1, 9, (64, 8)).astype(np.float64)           # every value non-zero and fp32-exact: auto keeps the dense format
    B.write_block(f, dense)
    assert B.read_header(f)["version"] == 1
    with open(f, "r+b") as fh:
        fh.seek(12); fh.write(b"\x07\x00\x00\x00")               # an unknown dtype code
    with pytest.raises(ValueError, match="bad header"):
        B.read_header(f)


def test_reference_file_order(monkeypatch):
    """order(as.numeric(gsub("\\D*([0-9]+).*$", "\\1", allfiles))) on FULL paths (R/SHARP_unlimited3.R:59-61)."""
    names = ["part_10.blk", "part_2.blk", "part_1.blk", "readme"]
    monkeypatch.setattr(B.os, "listdir", lambda d: list(names))
    monkeypatch.setattr(B.os.path, "isdir", lambda d: True)
    monkeypatch.setattr(B.os.path, "isfile", lambda f: True)
    got = [os.path.basename(f) for f in B.list_block_files("/data/cells/")]
    assert got == ["part_1.blk", "part_2.blk", "part_10.blk", "readme"]          # no digits -> NA -> last
    # a number in a parent directory is the FIRST number of every path: all keys tie and list.files() order (sorted) stays
    got = [os.path.basename(f) for f in B.list_block_files("/data/run7")]
    assert got == ["part_1.blk", "part_10.blk", "part_2.blk", "readme"]
    monkeypatch.setattr(B.os.path, "isdir", lambda d: False)
    with pytest.raises(FileNotFoundError, match="should be a folder"):
        B.list_block_files("/nope")
