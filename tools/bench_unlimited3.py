"""Measures SHARP_unlimited3 end to end from block files (disk -> pinned memory -> HBM, one block ahead of the clustering)
against SHARP_unlimited on the same blocks already resident in HBM.  Usage: python tools/bench_unlimited3.py [dir] [nblocks] [cells]"""
import os
import shutil
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import sharp_amd
from sharp_amd import blocks as B
from sharp_amd import device as dev

sharp_amd.init(0)
root = sys.argv[1] if len(sys.argv) > 1 else "/dev/shm"
nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 4
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 25000
m = 20000
d = os.path.join(root, "sharpblk_bench")
shutil.rmtree(d, ignore_errors=True)
os.makedirs(d)
try:
    x = torch.empty((nb, m), dtype=torch.float32, device="cuda")
    for b in range(nblk):                       # write the synthetic blocks in the block-file format straight from the device
        dev.synth_fill(x, 20261003, b * nb)
        hdr = B._HDR.pack(B.MAGIC, 1, 0, m, nb, m)
        with open(os.path.join(d, "part_%d.blk" % (b + 1)), "wb") as fh:
            fh.write(hdr)
            x.cpu().numpy().tofile(fh)
    del x
    nd = {"dir": d, "ncells": nblk * nb, "ngenes": m}
    for rep in range(2):
        t0 = time.perf_counter()
        res = sharp_amd.SHARP_unlimited3(nd, rN_seed=2103, viewflag=False)
        dt = time.perf_counter() - t0
        print("SHARP_unlimited3 run %d: %d cells in %.3f s = %.0f cells/s, %.2f GB streamed = %.2f GB/s incl. clustering"
              % (rep, nblk * nb, dt, nblk * nb / dt, res["bytes_streamed"] / 1e9, res["bytes_streamed"] / 1e9 / dt), flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
