"""Directory-of-blocks input for SHARP_unlimited3 (R/SHARP_unlimited3.R:29-235) and its host->HBM streaming.

The reference keeps each partition as an .rds file and `readRDS()`s them one by one (:103-104).  R is not needed to read
the format used here: a 64-byte header followed by the block exactly as it lives in HBM -- cells x ld float32, one cell
per row (genes x cells column-major in R's terms), ld = genes rounded up to 4 -- so a file goes page-locked buffer -> DMA
with no conversion.  An R maintainer writes it with `writeBin`.  While block i is clustered, block i+1
is read from disk into pinned memory by a worker thread and copied to the GPU on a side stream."""
import os
import re
import struct
import threading

import numpy as np

MAGIC = b"SHARPBLK"
_HDR = struct.Struct("<8sIIQQQ24x")          # magic, version, dtype (0 = float32), genes, cells, ld
HEADER_BYTES = 64
assert _HDR.size == HEADER_BYTES


def write_block(path, X):
    """X: (genes, cells) array-like (what one element of the reference's scExp list is)."""
    X = np.asarray(X)
    m, n = X.shape
    ld = (m + 3) // 4 * 4
    buf = np.zeros((n, ld), np.float32)
    buf[:, :m] = X.T
    with open(path, "wb") as fh:
        fh.write(_HDR.pack(MAGIC, 1, 0, m, n, ld))
        buf.tofile(fh)


def read_header(path):
    with open(path, "rb") as fh:
        raw = fh.read(HEADER_BYTES)
    if len(raw) != HEADER_BYTES:
        raise ValueError("%s: not a SHARP block file (short header)" % path)
    magic, ver, dtype, m, n, ld = _HDR.unpack(raw)
    if magic != MAGIC or ver != 1 or dtype != 0 or ld < m or ld % 4:
        raise ValueError("%s: not a SHARP block file (bad header)" % path)
    if os.path.getsize(path) != HEADER_BYTES + n * ld * 4:
        raise ValueError("%s: truncated block file" % path)
    return {"genes": int(m), "cells": int(n), "ld": int(ld)}


def read_block(path):
    """-> the (genes, cells) float32 array a block file holds (what write_block was given)."""
    h = read_header(path)
    buf = np.fromfile(path, dtype=np.float32, offset=HEADER_BYTES, count=h["cells"] * h["ld"]).reshape(h["cells"], h["ld"])
    return np.ascontiguousarray(buf[:, :h["genes"]].T)


def list_block_files(directory):
    """list.files() ordered by the first run of digits of each path (R/SHARP_unlimited3.R:59-61)."""
    d = directory[:-1] if directory.endswith("/") else directory
    if not os.path.isdir(d):
        raise FileNotFoundError("%s should be a folder storing several partitions of single-cell datasets!" % d)
    files = [os.path.join(d, f) for f in sorted(os.listdir(d)) if os.path.isfile(os.path.join(d, f))]

    def key(path):
        # as.numeric(gsub("\\D*([0-9]+).*$", "\\1", path)): leading non-digits dropped, the first digit run kept
        mt = re.match(r"\D*([0-9]+)", path)
        return float(mt.group(1)) if mt else float("inf")

    order = sorted(range(len(files)), key=lambda i: key(files[i]))   # order() is stable, like sorted()
    return [files[i] for i in order]


class BlockStreamer:
    """Iterates (index, header, device tensor (cells, genes) float32) over block files with one block of read-ahead.

    Two pinned host buffers and two device buffers; the worker thread reads file i+1 and enqueues its host->device copy on
    a side stream while the caller works on block i; the consumer's stream waits on the copy's event only."""

    def __init__(self, files, device="cuda", read_threads=8):
        import torch

        self.torch = torch
        self.files = list(files)
        self.hdrs = [read_header(f) for f in self.files]
        cap = max(h["cells"] * h["ld"] for h in self.hdrs) if self.hdrs else 0
        self.pinned = [torch.empty(cap, dtype=torch.float32).pin_memory() for _ in range(2)]
        self.dev = [torch.empty(cap, dtype=torch.float32, device=device) for _ in range(2)]
        self.copy_stream = torch.cuda.Stream()
        self.events = [None, None]
        self.threads = [None, None]
        self.bytes_streamed = 0
        self.read_threads = int(read_threads)

    def _load(self, i):
        slot = i & 1
        h = self.hdrs[i]
        cnt = h["cells"] * h["ld"]
        dst = memoryview(self.pinned[slot].numpy()[:cnt]).cast("B")
        nbytes = cnt * 4
        fd = os.open(self.files[i], os.O_RDONLY)
        try:
            # several positional reads in parallel (they release the GIL): one thread copies from the page cache at a few GB/s
            nthr = max(1, min(self.read_threads, nbytes >> 26))
            step = (nbytes + nthr - 1) // nthr
            step = (step + 4095) // 4096 * 4096
            errs = []

            def part(lo):
                hi = min(lo + step, nbytes)
                pos = lo
                while pos < hi:
                    got = os.preadv(fd, [dst[pos:hi]], HEADER_BYTES + pos)
                    if got <= 0:
                        errs.append("%s: short read" % self.files[i])
                        return
                    pos += got

            ths = [threading.Thread(target=part, args=(lo,), daemon=True) for lo in range(0, nbytes, step)]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            if errs:
                raise IOError(errs[0])
        finally:
            os.close(fd)
        torch = self.torch
        with torch.cuda.stream(self.copy_stream):
            self.dev[slot][:cnt].copy_(self.pinned[slot][:cnt], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        self.events[slot] = ev
        self.bytes_streamed += cnt * 4

    def _start(self, i):
        t = threading.Thread(target=self._load, args=(i,), daemon=True)
        t.start()
        self.threads[i & 1] = t

    def __iter__(self):
        n = len(self.files)
        if n:
            self._start(0)
        for i in range(n):
            slot = i & 1
            self.threads[slot].join()
            self.events[slot].synchronize()              # block i is resident
            if i + 1 < n:
                self._start(i + 1)                       # the other slot: its previous user (block i-1) is finished
            h = self.hdrs[i]
            yield i, h, self.dev[slot][: h["cells"] * h["ld"]].view(h["cells"], h["ld"])[:, : h["genes"]]
