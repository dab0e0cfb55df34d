import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sharp_amd
from sharp_amd import device as dev
sharp_amd.init(0)
n, m, K = 50000, 20000, 15
dX = torch.empty((n, m), dtype=torch.float32, device="cuda")
dev.synth_fill(dX, 20261003, 0, 12, 1000)
for it in range(6):
    t0 = time.perf_counter(); dev.SHARP_dev(dX, ensize_K=K, rN_seed=2103); print("back-to-back %.1f ms" % ((time.perf_counter() - t0) * 1e3))
for gap in (0.05, 0.2, 1.0):
    for it in range(3):
        time.sleep(gap)
        t0 = time.perf_counter(); dev.SHARP_dev(dX, ensize_K=K, rN_seed=2103); print("after %.2f s idle: %.1f ms" % (gap, (time.perf_counter() - t0) * 1e3))
dev.profile(True)
time.sleep(1.0)
dev.SHARP_dev(dX, ensize_K=K, rN_seed=2103)
tab = dev.profile_table()
print(sorted(((k, round(v[0], 2)) for k, v in tab.items()), key=lambda kv: -kv[1])[:14])
