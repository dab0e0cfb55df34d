"""Agglomeration of a FEW tasks at a time (the 'keep the working set in the Infinity Cache' idea): SHARP_dev on
n = 2000 * T cells with one projection = T base tasks of 2000 observations in the round-per-launch form (a task spread over
SHARP_HC_WPT workgroups); prints the agglomeration time per task.  usage: hc_mall.py T [T ...]"""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import sharp_amd
from sharp_amd import device as dev

sharp_amd.init(0)
m = 20000
for T in [int(a) for a in sys.argv[1:]] or [4, 8, 16, 32]:
    n = 2000 * T
    dX = torch.empty((n, m), dtype=torch.float32, device="cuda")
    dev.synth_fill(dX, 20261003, 0, 12, 1000)
    dev.SHARP_dev(dX, ensize_K=1, rN_seed=2103)
    dev.profile(True)
    reps = 3
    for _ in range(reps):
        dev.SHARP_dev(dX, ensize_K=1, rN_seed=2103)
    prof = dev.profile_table()
    h = prof.get("hclust", (0, 0))[0] / reps
    print("T=%d tasks: agglomeration %.2f ms = %.3f ms per task (375 tasks in chunks of 188, one launch each: 0.073 ms per task)" % (T, h, h / T), flush=True)
    dev.profile(False)
    del dX
