"""Per-kernel and host-section times of one SHARP_unlimited block step (sharp_unlimited_block_dev) at the K = 5 shapes:
cfg3 / cfg5 block (50 000 x 20 000, p = 474 / 582) and cfg4's per-GPU share (162 500 x 27 000, p = 508)."""
import json
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import sharp_amd
from sharp_amd import device as dev

sharp_amd.init(0)
lib = sharp_amd.lib()
which = sys.argv[1:] or ["cfg3", "cfg4"]
for tag, (n, m, p) in {"cfg3": (50000, 20000, 474), "cfg5": (50000, 20000, 582), "cfg4": (162500, 27000, 508)}.items():
    if tag not in which:
        continue
    x = torch.empty((n, m), dtype=torch.float32, device="cuda")
    dev.synth_fill(x, 20261003, 0)
    proj = sharp_amd.Projector(m, p, [50 + 2103 + k for k in range(1, 6)])
    dev.unlimited_block_dev(x, p, proj.handle, 5, 2103)
    dev.unlimited_block_dev(x, p, proj.handle, 5, 2103)
    dev.profile(True)
    reps = int(__import__("os").environ.get("BLOCK_REPS", "4"))
    lib.sharp_synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        dev.unlimited_block_dev(x, p, proj.handle, 5, 2103)
    lib.sharp_synchronize()
    dt = (time.perf_counter() - t0) / reps
    prof = dev.profile_table()
    dev.profile(False)
    print(json.dumps({"block": tag, "cells": n, "genes": m, "p": p, "ms_per_block": round(dt * 1e3, 2),
                      "ms": {k: round(v[0] / reps, 2) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0]) if v[0] / reps >= 0.05}}), flush=True)
    proj.close()
    del x
    torch.cuda.empty_cache()
