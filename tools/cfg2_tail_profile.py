"""Host timers of one pipelined cfg2 step (profiling on keeps the two-chunk pipeline; KernelTimer events add a little)."""
import sys, time, ctypes as C
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import sharp_amd
from sharp_amd import device as dev
sharp_amd.init(0)
n, m, K = 50000, 20000, 15
x = torch.empty((n, m), dtype=torch.float32, device="cuda"); dev.synth_fill(x, 20261003, 0, 25, 1000)
def call():
    return dev.SHARP_dev(x, ensize_K=K, rN_seed=2103)
for _ in range(3): call()
torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter(); call(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("plain steps ms", [round(t, 2) for t in ts])
dev.profile(True)
t0 = time.perf_counter(); call(); torch.cuda.synchronize(); t1 = time.perf_counter()
print("profiled call %.1f ms" % ((t1 - t0) * 1e3))
prof = dev.profile_table()
for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0]):
    print("%-34s %8.2f ms  x%d" % (k, v[0], v[1]))
