"""Host-inclusive SHARP_unlimited on a LIST of host blocks (what an R session holds): cfg3-shaped data (B blocks of 50 000 cells x 20 000
genes), dense fp64 matrices against dgCMatrix-like sparse blocks, on one device and on several device slots.  Prints seconds per call,
cells/s and the per-worker timeline (upload hidden under clustering).  usage: bench_host_blocks.py [blocks=4] [cells_per_block=50000]"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sharp_amd
from sharp_amd import device as dev

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
m, K = 20000, 5
sharp_amd.init(0)
dense, sparse, dblocks = [], [], []
for b in range(B):
    dX = torch.empty((nb, m), dtype=torch.float32, device="cuda")
    dev.synth_fill(dX, 20261003, b * nb, 12, 1000)
    X = np.asfortranarray(dX.cpu().numpy().T.astype(np.float64))
    dense.append(X)
    sparse.append(sp.csc_matrix(X))
    dblocks.append(dX)
torch.cuda.synchronize()
n = B * nb
print("%d blocks of %d cells x %d genes: dense %.1f GB fp64 on the host, sparse %.2f GB (nnz %.1f %%)"
      % (B, nb, m, sum(x.nbytes for x in dense) / 1e9, sum(s.nnz for s in sparse) * 12 / 1e9, 100.0 * sparse[0].nnz / dense[0].size), flush=True)


def timeline(tag):
    tl = dev.multi_timeline()
    print("   timeline of %s (worker block | upload start-end | clustering start-end, seconds):" % tag)
    for r in tl[np.lexsort((tl[:, 1], tl[:, 0]))]:
        print("      w%d b%d | %.3f-%.3f | %.3f-%.3f" % (r[0], r[1], r[2], r[3], r[4], r[5]))
    up = float((tl[:, 3] - tl[:, 2]).sum())
    wall = float(tl[:, 5].max())
    busy = float((tl[:, 5] - tl[:, 4]).sum())
    print("   sum of uploads %.3f s, sum of clustering %.3f s, last block done at %.3f s" % (up, busy, wall), flush=True)


ref = None
for name, blocks, devices in (("resident (sharp_SHARP_unlimited_multi_dev, 1 slot)", None, [0]),
                              ("dense host, 1 slot, pipelined upload", dense, [0]),
                              ("sparse host, 1 slot, pipelined upload", sparse, [0]),
                              ("sparse host, 2 slots on GPU 0", sparse, [0, 0]),
                              ("dense host, upload-all-then-cluster (sharp_SHARP_unlimited_view)", dense, None)):
    for it in range(2):
        t0 = time.perf_counter()
        if blocks is None:
            pred, npred, p, _ = dev.unlimited_multi_dev(dblocks, [0] * B, devices, ensize_K=K, rN_seed=2103)
        else:
            res = sharp_amd.SHARP_unlimited(blocks, ensize_K=K, rN_seed=2103, viewflag=False, devices=devices)
            pred = res["pred_clusters"]
        t = time.perf_counter() - t0
    ref = pred if ref is None else ref
    print("%-70s %.3f s = %.0f cells/s; labels identical to the resident run: %s" % (name, t, n / t, np.array_equal(pred, ref)), flush=True)
    if devices is not None:
        timeline(name)
