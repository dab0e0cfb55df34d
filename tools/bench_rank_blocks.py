"""A rank's 25 blocks of cfg5 (50 000 x 20 000, K = 5, p = 582): block after block (sharp_unlimited_block_dev, the next block prepared
under the current one's tail) against windows of ten resident blocks (sharp_unlimited_blocks_dev)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import sharp_amd
from sharp_amd import device as dev
sharp_amd.init(0)
nb, m, K, p, RN, SEED = 50000, 20000, 5, 582, 2103, 20261003
W = int(sys.argv[1]) if len(sys.argv) > 1 else 10
proj = sharp_amd.Projector(m, p, [50 + RN + k for k in range(1, K + 1)])
bufs = [torch.empty((nb, m), dtype=torch.float32, device="cuda") for _ in range(W)]
def fill(b0):
    for q, x in enumerate(bufs): dev.synth_fill(x, SEED, (b0 + q) * nb, 12, 1000)
    torch.cuda.synchronize()
for rep in range(2):
    t_one = t_many = 0.0
    tabs_one, tabs_many = [], []
    for b0 in range(0, 20, W):
        fill(b0)
        t0 = time.perf_counter()
        for q, x in enumerate(bufs): tabs_one.append(dev.unlimited_block_dev(x, p, proj.handle, K, RN, next_block=bufs[q + 1] if q + 1 < W else None))
        t_one += time.perf_counter() - t0
        t0 = time.perf_counter()
        tabs_many += dev.unlimited_blocks_dev(bufs, p, proj.handle, K, RN)
        t_many += time.perf_counter() - t0
    same = all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(tabs_one, tabs_many))
    print("20 blocks: block after block %.1f ms per block, windows of %d %.1f ms per block, identical tables: %s" % (t_one / 20 * 1e3, W, t_many / 20 * 1e3, same))
proj.close()
