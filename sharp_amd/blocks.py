"""Directory-of-blocks input for SHARP_unlimited3 (R/SHARP_unlimited3.R:29-235) and its host->HBM streaming.

The reference keeps each partition as an .rds file and `readRDS()`s them one by one (:103-104).  R is not needed to read the two
formats used here (an R maintainer writes them with `writeBin`); both start with a 64-byte header:

  version 1, dense:  the block exactly as it lives in HBM -- cells x ld float32, one cell per row (genes x cells column-major in R's
             terms), ld = genes rounded up to 4 -- so a file goes page-locked buffer -> DMA with no conversion: 4 GB per 50 000 x 20 000 block;
  version 2, packed: the block as the three slots of a dgCMatrix in their narrowest exact types -- column pointers (cells + 1 int64), row
             indices (16 bits up to 65 536 genes, else 32), values (unsigned 8- or 16-bit integers for counts, else float, else double) --
             3 bytes per non-zero for typical count data: 0.35 GB for the same block.  The bytes go page-locked buffer -> DMA as they are and
             the dense block is built on the device (sharp_csc_packed_expand_dev); values that fp32 cannot hold give an fp64 block.

While earlier blocks are clustered a reader thread brings the next files into a ring of pinned buffers and enqueues their copies on a
side stream; the consumer takes the blocks that have arrived -- several at a time when the clustering is the slower side, as one
pipelined batch (SHARP_unlimited3)."""
import ctypes as C
import os
import re
import struct
import threading

import numpy as np

MAGIC = b"SHARPBLK"
_HDR = struct.Struct("<8sIIQQQ24x")          # version 1: magic, version, dtype (0 = float32), genes, cells, ld
_HDR2 = struct.Struct("<8sIIQQQII16x")       # version 2: magic, version, dtype (1 = packed CSC), genes, cells, nnz, index bits, value bits
HEADER_BYTES = 64
assert _HDR.size == HEADER_BYTES and _HDR2.size == HEADER_BYTES


def _pad16(nbytes):
    return (nbytes + 15) // 16 * 16


def _packed_layout(n, nnz, idx_bits, val_bits):
    """byte offsets (from the end of the header) of the three arrays of a version-2 file, and the payload size"""
    o_idx = _pad16((n + 1) * 8)
    o_val = o_idx + _pad16(nnz * idx_bits // 8)
    return o_idx, o_val, o_val + _pad16(nnz * val_bits // 8)


def write_block(path, X, fmt="auto"):
    """X: (genes, cells) array-like or scipy.sparse matrix (what one element of the reference's scExp list is).
    fmt: "dense" (version 1, float32), "packed" (version 2), "auto": packed unless more than 30 % of the values are non-zero and every value
    survives float32."""
    sparse = hasattr(X, "tocsc")
    if sparse:
        Xc = X.tocsc()
        Xc.sum_duplicates()
        Xc.sort_indices()
        m, n = Xc.shape
        vals, nnz = np.asarray(Xc.data, np.float64), int(Xc.nnz)
    else:
        X = np.asarray(X)
        m, n = X.shape
        nnz = int(np.count_nonzero(X))
        vals = None
    if fmt == "auto":
        f32_ok = bool(np.all(np.asarray(X if not sparse else vals, np.float64).astype(np.float32).astype(np.float64) == np.asarray(X if not sparse else vals, np.float64)))
        fmt = "dense" if (not sparse and nnz > 0.3 * m * n and f32_ok) else "packed"
    if fmt == "dense":
        Xd = np.asarray(X.toarray() if sparse else X)
        ld = (m + 3) // 4 * 4
        buf = np.zeros((n, ld), np.float32)
        buf[:, :m] = Xd.T
        with open(path, "wb") as fh:
            fh.write(_HDR.pack(MAGIC, 1, 0, m, n, ld))
            buf.tofile(fh)
        return
    if not sparse:
        import scipy.sparse as sps

        Xc = sps.csc_matrix(X)
        Xc.sort_indices()
        vals = np.asarray(Xc.data, np.float64)
    idx_bits = 16 if m <= 65536 else 32
    if vals.size == 0 or (np.all(vals >= 0) and np.all(vals <= 255) and np.all(vals == np.floor(vals))):
        val_bits, v = 8, vals.astype(np.uint8)
    elif np.all(vals >= 0) and np.all(vals <= 65535) and np.all(vals == np.floor(vals)):
        val_bits, v = 16, vals.astype(np.uint16)
    elif np.all(vals.astype(np.float32).astype(np.float64) == vals):
        val_bits, v = 32, vals.astype(np.float32)
    else:
        val_bits, v = 64, vals
    o_idx, o_val, total = _packed_layout(n, nnz, idx_bits, val_bits)
    buf = np.zeros(total, np.uint8)
    buf[: (n + 1) * 8] = np.asarray(Xc.indptr, np.int64).view(np.uint8)
    ib = np.asarray(Xc.indices).astype(np.uint16 if idx_bits == 16 else np.int32)
    buf[o_idx: o_idx + ib.nbytes] = ib.view(np.uint8)
    buf[o_val: o_val + v.nbytes] = v.view(np.uint8)
    with open(path, "wb") as fh:
        fh.write(_HDR2.pack(MAGIC, 2, 1, m, n, nnz, idx_bits, val_bits))
        buf.tofile(fh)


def read_header(path):
    with open(path, "rb") as fh:
        raw = fh.read(HEADER_BYTES)
    if len(raw) != HEADER_BYTES:
        raise ValueError("%s: not a SHARP block file (short header)" % path)
    magic, ver, dtype, m, n, third = _HDR.unpack(raw)
    if magic != MAGIC:
        raise ValueError("%s: not a SHARP block file (bad header)" % path)
    if ver == 1:
        ld = third
        if dtype != 0 or ld < m or ld % 4:
            raise ValueError("%s: not a SHARP block file (bad header)" % path)
        if os.path.getsize(path) != HEADER_BYTES + n * ld * 4:
            raise ValueError("%s: truncated block file" % path)
        return {"version": 1, "genes": int(m), "cells": int(n), "ld": int(ld), "f64": False, "payload": int(n * ld * 4)}
    if ver == 2:
        _, _, _, m, n, nnz, idx_bits, val_bits = _HDR2.unpack(raw)
        if dtype != 1 or idx_bits not in (16, 32) or val_bits not in (8, 16, 32, 64) or (idx_bits == 16 and m > 65536):
            raise ValueError("%s: not a SHARP block file (bad header)" % path)
        o_idx, o_val, total = _packed_layout(n, nnz, idx_bits, val_bits)
        if os.path.getsize(path) != HEADER_BYTES + total:
            raise ValueError("%s: truncated block file" % path)
        f64 = val_bits == 64
        return {"version": 2, "genes": int(m), "cells": int(n), "ld": int((m + 1) // 2 * 2 if f64 else (m + 3) // 4 * 4), "f64": f64,
                "nnz": int(nnz), "idx_bits": int(idx_bits), "val_bits": int(val_bits), "o_idx": o_idx, "o_val": o_val, "payload": int(total)}
    raise ValueError("%s: not a SHARP block file (version %d)" % (path, ver))


def read_block(path):
    """-> the (genes, cells) array a block file holds (float32, or float64 for a packed file with 64-bit values)"""
    h = read_header(path)
    if h["version"] == 1:
        buf = np.fromfile(path, dtype=np.float32, offset=HEADER_BYTES, count=h["cells"] * h["ld"]).reshape(h["cells"], h["ld"])
        return np.ascontiguousarray(buf[:, :h["genes"]].T)
    raw = np.fromfile(path, dtype=np.uint8, offset=HEADER_BYTES, count=h["payload"])
    n, nnz = h["cells"], h["nnz"]
    cp = raw[: (n + 1) * 8].view(np.int64)
    idx = raw[h["o_idx"]: h["o_idx"] + nnz * h["idx_bits"] // 8].view(np.uint16 if h["idx_bits"] == 16 else np.int32)
    val = raw[h["o_val"]: h["o_val"] + nnz * h["val_bits"] // 8].view({8: np.uint8, 16: np.uint16, 32: np.float32, 64: np.float64}[h["val_bits"]])
    out = np.zeros((h["genes"], n), np.float64 if h["f64"] else np.float32)
    cols = np.repeat(np.arange(n), np.diff(cp))
    out[idx.astype(np.int64), cols] = val
    return out


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


# The buffers of the last streamer that was closed, for the next one of the same shape: page-locking eight file-sized buffers costs a few
# tenths of a second per call (0.37 s of a 0.9 s call over twenty block files); sharp_amd.blocks.release_buffers() drops them.
_SPARE = {}


def release_buffers():
    _SPARE.clear()


class BlockStreamer:
    """Block files -> device tensors (cells, genes), float32 or float64, through a ring of buffers.

    A reader thread takes the files in order: positional reads (several threads) into a pinned buffer of the ring, then the copy to the
    device on a side stream -- a dense file straight into its block, a packed file as it is into a staging buffer.  The consumer waits for
    the copy's event only; a packed block is expanded into its dense block on the library's stream (sharp_csc_packed_expand_dev) when the
    consumer takes it.  `for i, hdr, dX in streamer` yields one block at a time; `streamer.groups(g)` yields lists of up to g blocks: those
    that have ARRIVED when the consumer asks (at least one: it waits for the first).  A block's buffers are reused once the consumer has
    asked for what comes after it.  Ring: eight buffers (files up to 1 GB), two for larger files."""

    def __init__(self, files, device="cuda", read_threads=8, ring=None):
        import torch

        self.torch = torch
        self.files = list(files)
        self.hdrs = [read_header(f) for f in self.files]
        pay = max((h["payload"] for h in self.hdrs), default=0)
        self.ring = int(ring) if ring else (8 if pay <= (1 << 30) else 2)
        dense_bytes = max((h["cells"] * h["ld"] * (8 if h["f64"] else 4) for h in self.hdrs), default=0)
        any_packed = any(h["version"] == 2 for h in self.hdrs)
        self._key = (str(device), self.ring)
        spare = _SPARE.pop(self._key, None)
        need = (max(pay, 16), max(pay, 16) if any_packed else 16, max(dense_bytes, 16))
        if spare is not None and all(spare[k][0].numel() >= need[k] for k in range(3)):
            self.pinned, self.stage, self.dense = spare
        else:
            del spare
            self.pinned = [torch.empty(need[0], dtype=torch.uint8).pin_memory() for _ in range(self.ring)]
            self.stage = [torch.empty(need[1], dtype=torch.uint8, device=device) for _ in range(self.ring)]
            self.dense = [torch.empty(need[2], dtype=torch.uint8, device=device) for _ in range(self.ring)]
        self.copy_stream = torch.cuda.Stream()
        self.events = [None] * len(self.files)
        self.bytes_streamed = 0
        self.read_threads = int(read_threads)
        self.read_seconds = 0.0                       # wall time the reader thread spent reading files (hidden under the clustering or not)
        self.wait_seconds = 0.0                       # wall time the CONSUMER waited for a block that had not arrived
        self.expand_seconds = 0.0                     # wall time the consumer spent turning packed blocks into dense ones (its own stream, before the clustering)
        self._mu = threading.Condition()
        self._ready = 0                               # files [0, _ready) are on their way to the device (event recorded)
        self._consumed = 0                            # blocks [0, _consumed) are finished with
        self._err = None
        self._stop = False
        self._thread = None

    # ---- reader thread
    def _read_file(self, i, slot):
        h = self.hdrs[i]
        nbytes = h["payload"]
        dst = memoryview(self.pinned[slot].numpy()[:nbytes]).cast("B")
        fd = os.open(self.files[i], os.O_RDONLY)
        try:
            # several positional reads in parallel (they release the GIL): one thread copies from the page cache at a few GB/s
            nthr = max(1, min(self.read_threads, nbytes >> 24))
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

    def _reader(self):
        import time

        torch = self.torch
        try:
            for i in range(len(self.files)):
                with self._mu:
                    self._mu.wait_for(lambda: self._stop or self._consumed + self.ring > i)
                    if self._stop:
                        return
                slot = i % self.ring
                t0 = time.perf_counter()
                self._read_file(i, slot)
                self.read_seconds += time.perf_counter() - t0
                h = self.hdrs[i]
                nbytes = h["payload"]
                with torch.cuda.stream(self.copy_stream):
                    target = self.dense[slot] if h["version"] == 1 else self.stage[slot]
                    target[:nbytes].copy_(self.pinned[slot][:nbytes], non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(self.copy_stream)
                self.events[i] = ev
                self.bytes_streamed += nbytes
                with self._mu:
                    self._ready = i + 1
                    self._mu.notify_all()
        except BaseException as e:  # noqa: BLE001
            with self._mu:
                self._err = e
                self._mu.notify_all()

    def _ensure_started(self):
        if self._thread is None:
            self._thread = threading.Thread(target=self._reader, daemon=True)
            self._thread.start()

    def close(self):
        with self._mu:
            self._stop = True
            self._mu.notify_all()
        if self._thread is not None:
            self._thread.join()
            self._thread = None
        if self.pinned is not None:                       # (the reader is gone and the consumer is done with the blocks: the buffers may serve the next streamer)
            self.torch.cuda.synchronize()
            _SPARE[self._key] = (self.pinned, self.stage, self.dense)
            self.pinned = self.stage = self.dense = None

    # ---- consumer side
    def _take(self, i):
        """block i as a device tensor (its copy has been enqueued); a packed block is expanded now, on the library's stream"""
        from ._lib import check, lib

        import time

        torch = self.torch
        h = self.hdrs[i]
        slot = i % self.ring
        t_ex = time.perf_counter()
        self.events[i].synchronize()
        dt = torch.float64 if h["f64"] else torch.float32
        n, ld, m = h["cells"], h["ld"], h["genes"]
        dX = self.dense[slot][: n * ld * (8 if h["f64"] else 4)].view(dt).view(n, ld)
        if h["version"] == 2:
            base = self.stage[slot].data_ptr()
            check(lib().sharp_csc_packed_expand_dev(C.c_void_p(base), C.c_void_p(base + h["o_idx"]), h["idx_bits"], C.c_void_p(base + h["o_val"]),
                                                    h["val_bits"], m, C.c_longlong(n), C.c_void_p(dX.data_ptr()), C.c_longlong(ld), int(h["f64"])))
        self.expand_seconds += time.perf_counter() - t_ex
        return dX[:, :m]

    def groups(self, max_group=3):
        """yields lists of (index, header, device tensor): the blocks that have arrived, at most max_group (and at most ring - 1) at a time"""
        import time

        self._ensure_started()
        n = len(self.files)
        g = max(1, min(int(max_group), self.ring - 1))
        i = 0
        try:
            while i < n:
                with self._mu:
                    self._consumed = i                         # everything before block i is finished with
                    self._mu.notify_all()
                    t0 = time.perf_counter()
                    self._mu.wait_for(lambda: self._err is not None or self._ready > i)
                    self.wait_seconds += time.perf_counter() - t0
                    if self._err is not None:
                        raise self._err
                    j = min(self._ready, i + g, n)
                yield [(q, self.hdrs[q], self._take(q)) for q in range(i, j)]
                i = j
        finally:
            self.close()

    def __iter__(self):
        for grp in self.groups(1):
            yield grp[0]
