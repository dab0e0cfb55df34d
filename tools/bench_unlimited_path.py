"""What one rank of `bench.py --gpus N` does per step (projector, one SHARP_unlimited block, centroid merge) against the N = 1 step
(one SHARP() call), on one GPU with the same data: the per-rank overhead that bounds the weak-scaling efficiency before RCCL."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sharp_amd
from sharp_amd import device as dev

sharp_amd.init(0)
n, m, K, RS = 50000, 20000, 15, 2103
dX = torch.empty((n, m), dtype=torch.float32, device="cuda")
dev.synth_fill(dX, 20261003, 0, 12, 1000)


def step_single():
    dev.SHARP_dev(dX, ensize_K=K, rN_seed=RS)


def timeit(f, reps=5):
    f(); f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    sharp_amd.lib().sharp_synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


print("N=1 step (SHARP_dev, p=391): %.2f ms" % timeit(step_single))
import math

for w in (1, 2, 4, 8):
    p = int(math.ceil(math.log2(n * w) / 0.04))

    def f():
        proj = sharp_amd.Projector(m, p, [50 + RS + k for k in range(1, K + 1)])
        pr, mn, cn = dev.unlimited_block_dev(dX, p, proj.handle, K, RS)
        dev.unlimited_merge(mn, cn, n)
        proj.close()
    print("rank step of a %d-GPU run (p=%d): %.2f ms" % (w, p, timeit(f)))
