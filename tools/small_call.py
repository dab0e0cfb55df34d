"""A cfg1-sized SHARP() call (479 cells x 20 000 genes, K = 15) in a loop: for rocprofv3 --kernel-trace --stats."""
import sys, time, os
sys.path.insert(0, "/root/repo")
import torch, sharp_amd
from sharp_amd import device as dev
sharp_amd.init(0); lib = sharp_amd.lib()
x = torch.empty((479, 20000), dtype=torch.float32, device="cuda"); dev.synth_fill(x, 20261003, 0, 5, 1000)
for _ in range(3): dev.SHARP_dev(x, ensize_K=15, rN_seed=2103)
lib.sharp_synchronize(); t0 = time.perf_counter()
for _ in range(20): dev.SHARP_dev(x, ensize_K=15, rN_seed=2103)
lib.sharp_synchronize(); print("479 cells: %.3f ms per call" % ((time.perf_counter() - t0) / 20 * 1e3))
