"""Host-side logic of the SHARP_unlimited3 block files (no GPU): format round trip, validation, reference file order."""
import os

import numpy as np
import pytest

from sharp_amd import blocks as B


def test_block_file_round_trip(tmp_path):
    rng = np.random.default_rng(3)
    X = rng.poisson(0.3, size=(1003, 17)).astype(np.float64)       # genes x cells, genes not a multiple of 4
    f = str(tmp_path / "a.blk")
    B.write_block(f, X)
    h = B.read_header(f)
    assert h == {"genes": 1003, "cells": 17, "ld": 1004}
    raw = np.fromfile(f, np.float32, offset=B.HEADER_BYTES).reshape(17, 1004)
    assert np.array_equal(raw[:, :1003], X.T.astype(np.float32)) and np.all(raw[:, 1003:] == 0)
    with open(f, "r+b") as fh:                                       # truncated payload
        fh.truncate(os.path.getsize(f) - 4)
    with pytest.raises(ValueError, match="truncated"):
        B.read_header(f)
    g = str(tmp_path / "b.blk")
    open(g, "wb").write(b"not a block file" * 8)
    with pytest.raises(ValueError, match="bad header"):
        B.read_header(g)


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
