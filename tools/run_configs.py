"""Runs the larger BASELINE.json configurations on one GPU (synthetic blocks generated on the device):
  cfg3: 500 000 cells x 20 000 genes as 10 blocks x 50 000, SHARP_unlimited, K = 5
  cfg4 share: one 162 500-cell x 27 000-gene block of the 1.3 M-cell run (what each of 8 GPUs processes), p = 508
Prints wall time, cells/s, per-kernel milliseconds and ARI vs the planted clusters."""
import ctypes as C
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import sharp_amd
from sharp_amd import device as dev
from sharp_amd.api import ARI

sharp_amd.init(0)
lib = sharp_amd.lib()
SEED, RN = 20261003, 2103
which = sys.argv[1:] or ["cfg3", "cfg4share"]


def run_unlimited(blocks, cell0s, tag):
    m = blocks[0].shape[1]
    ncb = np.array([b.shape[0] for b in blocks], np.int64)
    ldb = np.array([b.stride(0) for b in blocks], np.int64)
    ptrs = (C.c_void_p * len(blocks))(*[b.data_ptr() for b in blocks])
    n = int(ncb.sum())
    pred = np.zeros(n, np.int32)
    npred, pu = C.c_int(), C.c_int()
    torch.cuda.synchronize()
    for attempt in ("first call (workspaces allocated)", "steady state"):
        dev.profile(True)
        t0 = time.perf_counter()
        rc = lib.sharp_SHARP_unlimited_dev(ptrs, ncb.ctypes.data_as(C.POINTER(C.c_longlong)), ldb.ctypes.data_as(C.POINTER(C.c_longlong)),
                                           len(blocks), m, 5, 0, 0, 0, C.c_double(RN), pred.ctypes.data_as(C.POINTER(C.c_int)),
                                           C.byref(npred), C.byref(pu))
        lib.sharp_synchronize()
        dt = time.perf_counter() - t0
        assert rc in (0, 16, 32, 48), lib.sharp_last_error()
        truth = np.concatenate([dev.synth_labels(SEED, c0, int(nb), 12) for c0, nb in zip(cell0s, ncb)])
        prof = dev.profile_table()
        top = {k: round(v[0], 1) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:8]}
        print(json.dumps({"config": tag, "run": attempt, "cells": n, "genes": m, "p": pu.value, "clusters": npred.value, "seconds": round(dt, 3),
                          "cells_per_s": round(n / dt, 1), "ari_vs_truth": round(float(ARI(truth, pred)["HA"]), 4), "top_ms": top}),
              flush=True)


if "cfg3" in which:
    m, nb, B = 20000, 50000, 10
    blocks = []
    for b in range(B):
        x = torch.empty((nb, m), dtype=torch.float32, device="cuda")
        dev.synth_fill(x, SEED, b * nb)
        blocks.append(x)
    run_unlimited(blocks, [b * nb for b in range(B)], "cfg3: 500k x 20k, 10 blocks, SHARP_unlimited K=5, 1 GPU")
    del blocks
    torch.cuda.empty_cache()

if "cfg4share" in which:
    # one GPU's share of cfg4 run as a 2-block unlimited call so that p follows the 1.3 M-cell rule is not possible on one GPU;
    # instead time a single 162 500 x 27 000 block through SHARP() with reduced.ndim = 508 and K = 5 (what unlimited_block does)
    m, nb = 27000, 162500
    x = torch.empty((nb, m), dtype=torch.float32, device="cuda")
    dev.synth_fill(x, SEED, 0)
    torch.cuda.synchronize()
    for attempt in ("first call (workspaces allocated)", "steady state"):
      dev.profile(True)
      t0 = time.perf_counter()
      pred, info = dev.SHARP_dev(x, ensize_K=5, reduced_ndim=508, rN_seed=RN)
      lib.sharp_synchronize()
      dt = time.perf_counter() - t0
      truth = dev.synth_labels(SEED, 0, nb, 12)
      prof = dev.profile_table()
      top = {k: round(v[0], 1) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:8]}
      print(json.dumps({"config": "cfg4 per-GPU share: 162 500 x 27 000 block, K=5, p=508", "run": attempt, "cells": nb, "genes": m, "clusters": info["N.pred_cluster"],
                      "seconds": round(dt, 3), "cells_per_s": round(nb / dt, 1),
                      "ari_vs_truth": round(float(ARI(truth, pred)["HA"]), 4), "top_ms": top}), flush=True)
