import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sharp_amd
from sharp_amd import device as dev
mode = sys.argv[1]
sharp_amd.init(0)
n, m, K = 50000, 20000, 15
dX = torch.empty((n, m), dtype=torch.float32, device="cuda")
dev.synth_fill(dX, 20261003, 0, 12, 1000)
if mode == "hostarray":
    big = np.ones((m, n))
if mode == "cpu":
    h = dX.cpu()
if mode == "cpu_small":
    h = dX[:10].cpu()
if mode == "first_then_cpu":
    dev.SHARP_dev(dX, ensize_K=K, rN_seed=2103)
    h = dX.cpu()
for it in range(4):
    dev.profile(True)
    t0 = time.perf_counter(); dev.SHARP_dev(dX, ensize_K=K, rN_seed=2103); t = time.perf_counter() - t0
    tab = dev.profile_table()
    print(mode, "%.1f ms" % (t * 1e3), "hclust %.1f gemm %.1f" % (tab["hclust"][0], tab["corr_dist_gemm"][0]), flush=True)
