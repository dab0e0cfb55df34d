"""Host-inclusive SHARP_unlimited on a LIST of host blocks (what an R session holds): cfg3-shaped data (B blocks of 50 000 cells x 20 000
genes), dense fp64 matrices against dgCMatrix-like sparse blocks, on one device and on several device slots.  Prints seconds per call,
cells/s, the bytes that crossed PCIe and their rate, and the per-worker timeline (upload hidden under clustering).
usage: bench_host_blocks.py [blocks=10] [cells_per_block=50000] [dense_blocks=blocks]     (dense_blocks: how many of the blocks the dense runs
use -- ten dense fp64 blocks are 80 GB of host memory; fewer are taken when the box has less than twice that free)"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sharp_amd
from sharp_amd import device as dev

B = int(sys.argv[1]) if len(sys.argv) > 1 else 10
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
m, K = 20000, 5
BD = int(sys.argv[3]) if len(sys.argv) > 3 else B
avail = 0
for ln in open("/proc/meminfo"):
    if ln.startswith("MemAvailable"):
        avail = int(ln.split()[1]) * 1024
while BD > 2 and BD * nb * m * 8 * 2 > avail:
    BD -= 1
print("host memory available %.0f GB: dense runs on %d of %d blocks" % (avail / 1e9, BD, B), flush=True)
sharp_amd.init(0)
dense, sparse, dblocks = [], [], []
for b in range(B):
    dX = torch.empty((nb, m), dtype=torch.float32, device="cuda")
    dev.synth_fill(dX, 20261003, b * nb, 12, 1000)
    X = np.asfortranarray(dX.cpu().numpy().T.astype(np.float64))
    if b < BD:
        dense.append(X)
    sparse.append(sp.csc_matrix(X))
    dblocks.append(dX)
torch.cuda.synchronize()
print("%d blocks of %d cells x %d genes: dense %.1f GB fp64 on the host (%d blocks), sparse %.2f GB as R holds it (12 B per non-zero; nnz %.1f %%)"
      % (B, nb, m, sum(x.nbytes for x in dense) / 1e9, BD, sum(s.nnz for s in sparse) * 12 / 1e9, 100.0 * sparse[0].nnz / dense[0].size), flush=True)
nnz = sum(s.nnz for s in sparse)


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
for name, blocks, devices, group in (("resident (sharp_SHARP_unlimited_multi_dev, 1 slot)", None, [0], None),
                                     ("sparse host, block after block (the default)", sparse, None, None),
                                     ("sparse host, groups of up to 3 (SHARP_HOST_GROUP=3)", sparse, None, "3"),
                                     ("sparse host, groups of up to 2 (SHARP_HOST_GROUP=2)", sparse, None, "2"),
                                     ("dense host, block after block (the default)", dense, None, None),
                                     ("dense host, groups of up to 3 (SHARP_HOST_GROUP=3)", dense, None, "3"),
                                     ("sparse host, 2 slots on GPU 0", sparse, [0, 0], None)):
    if group is None:
        os.environ.pop("SHARP_HOST_GROUP", None)
    else:
        os.environ["SHARP_HOST_GROUP"] = group
    sharp_amd.reload_options()
    nblk = B if blocks is None or blocks is sparse else BD
    n = nblk * nb
    ts = []
    for it in range(3):
        t0 = time.perf_counter()
        if blocks is None:
            pred, npred, p, _ = dev.unlimited_multi_dev(dblocks, [0] * B, devices, ensize_K=K, rN_seed=2103)
        else:
            res = sharp_amd.SHARP_unlimited(blocks, ensize_K=K, rN_seed=2103, viewflag=False, devices=devices)
            pred = res["pred_clusters"]
        ts.append(time.perf_counter() - t0)
    t = min(ts[1:])
    if ref is None:
        ref = pred
    same = np.array_equal(pred, ref) if nblk == B else "n/a (fewer blocks: another p)"
    wire = sharp_amd.lib().sharp_x_wire()
    sent = 0 if blocks is None else (nnz * (wire // 8 + 2) if blocks is sparse else n * m * (wire // 8))
    print("%-62s %.3f s = %.0f cells/s (calls: %s); wire %d-bit values, %.2f GB over PCIe = %.1f GB/s of the call; labels identical to the resident run: %s"
          % (name, t, n / t, " ".join("%.3f" % x for x in ts), wire, sent / 1e9, sent / 1e9 / t, same), flush=True)
    timeline(name)
os.environ.pop("SHARP_HOST_GROUP", None)
