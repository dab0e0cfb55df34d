"""SHARP_unlimited3 end to end from block FILES (disk / tmpfs -> pinned memory -> HBM through a ring of buffers, ahead of the clustering): cfg5's shape,
NBLK partitions of 50 000 cells x 20 000 genes in the packed format (3 B per non-zero), p from the total cell count as the reference computes it
(R/SHARP_unlimited3.R:66), against the same run block after block.  usage: python tools/bench_unlimited3.py [dir=/dev/shm] [nblocks=200] [cells=50000]"""
import os
import random
import shutil
import string
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import sharp_amd
from sharp_amd import blocks as B
from sharp_amd import device as dev

sharp_amd.init(0)
root = sys.argv[1] if len(sys.argv) > 1 else "/dev/shm"
nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 200
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 50000
m = 20000
d = os.path.join(root, "sharpblk_" + "".join(random.choice(string.ascii_lowercase) for _ in range(12)))     # (no digit in the path: R/SHARP_unlimited3.R:59-61)
os.mkdir(d)
try:
    x = torch.empty((nb, m), dtype=torch.float32, device="cuda")
    total, truth = 0, []
    t0 = time.perf_counter()
    for b in range(nblk):
        dev.synth_fill(x, bench.DATA_SEED, b * nb, bench.G_TRUE, bench.N_MARK)
        total += bench.write_packed_block_from_device(os.path.join(d, "part_%d.blk" % (b + 1)), x, B)
        truth.append(dev.synth_labels(bench.DATA_SEED, b * nb, nb, bench.G_TRUE))
    del x
    print("%d block files of %d cells x %d genes written in %.0f s: %.1f GB (%.2f GB each; the same blocks as dense float32 files: %.0f GB)"
          % (nblk, nb, m, time.perf_counter() - t0, total / 1e9, total / 1e9 / nblk, nblk * nb * m * 4 / 1e9), flush=True)
    nd = {"dir": d, "ncells": nblk * nb, "ngenes": m}
    ref = None
    for name, grp in (("arrived blocks together, up to 6 (default)", 6), ("again", 6), ("block after block", 1)):
        t0 = time.perf_counter()
        res = sharp_amd.SHARP_unlimited3(nd, rN_seed=2103, viewflag=False, group=grp)
        dt = time.perf_counter() - t0
        ref = res["pred_clusters"] if ref is None else ref
        print("%-44s %d cells in %.3f s = %.0f cells/s = %.1f blocks/s; files %.2f GB/s; reader %.2f s of reads, the clustering waited %.3f s for blocks (%.0f %% of the "
              "reading hidden), expansion %.2f s, clustering calls %.2f s; p = %d, %d clusters, ARI vs planted truth %.4f; labels equal to the first run: %s"
              % (name, nblk * nb, dt, nblk * nb / dt, nblk / dt, total / 1e9 / dt, res["read_seconds"], res["wait_seconds"],
                 100.0 * (1.0 - res["wait_seconds"] / max(res["read_seconds"], 1e-9)), res["expand_seconds"], res["cluster_seconds"], res["reduced.dim"], res["N.pred_clusters"],
                 float(sharp_amd.ARI(np.concatenate(truth), res["pred_clusters"])["HA"]), bool(np.array_equal(ref, res["pred_clusters"]))), flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
