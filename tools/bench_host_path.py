"""Host-buffer entry point (what the R glue binds): sharp_SHARP on a pageable fp64 genes x cells matrix, PCIe-inclusive."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sharp_amd
from sharp_amd import device as dev

sharp_amd.init(0)
n, m, K = int(sys.argv[1]) if len(sys.argv) > 1 else 50000, 20000, 15
dX = torch.empty((n, m), dtype=torch.float32, device="cuda")
dev.synth_fill(dX, 20261003, 0, 12, 1000)
X = dX.cpu().numpy().T.astype(np.float64)          # (genes, cells) fp64, as as.matrix() hands it over
X = np.asfortranarray(X)
print("host matrix %.1f GB fp64, host cores %d" % (X.nbytes / 1e9, os.cpu_count()), flush=True)
ref, _ = dev.SHARP_dev(dX, ensize_K=K, rN_seed=2103)
for it in range(3):
    dev.profile(True)
    t0 = time.perf_counter()
    res = sharp_amd.SHARP(X, ensize_K=K, rN_seed=2103, logflag=False, forview=False)
    t = time.perf_counter() - t0
    t0 = time.perf_counter()
    pred = sharp_amd.api._run_sharp(X, K, 0, 0, 0, None, 0, 0, 0, 0, 0, None, None, True, None, 2103, False)[0]
    t2 = time.perf_counter() - t0
    print("SHARP(host fp64): python front door %.3f s = %.0f cells/s; sharp_SHARP C call alone %.3f s = %.0f cells/s; identical to resident run: %s %s"
          % (t, n / t, t2, n / t2, np.array_equal(res["pred_clusters"], ref), np.array_equal(pred, ref)), flush=True)
    print("   ", {k: v for k, v in dev.profile_table().items() if k.startswith("host:upload")})
t0 = time.perf_counter()
ref, _ = dev.SHARP_dev(dX, ensize_K=K, rN_seed=2103)
print("resident: %.3f s" % (time.perf_counter() - t0))
import scipy.sparse as sp

S = sp.csc_matrix(X)
print("sparse: nnz %.1f M (%.1f %%), %.2f GB of (int32, fp64) pairs on the host" % (S.nnz / 1e6, 100.0 * S.nnz / X.size, S.nnz * 12 / 1e9), flush=True)
for it in range(3):
    dev.profile(True)
    t0 = time.perf_counter()
    pred = sharp_amd.api._run_sharp(S, K, 0, 0, 0, None, 0, 0, 0, 0, 0, None, None, True, None, 2103, False)[0]
    t2 = time.perf_counter() - t0
    print("sharp_SHARP_csc: %.3f s = %.0f cells/s; identical: %s" % (t2, n / t2, np.array_equal(pred, ref)),
          {k: v for k, v in dev.profile_table().items() if k.startswith("host:upload")}, flush=True)
for it in range(3):
    dev.profile(True)
    t0 = time.perf_counter()
    dev.SHARP_dev(dX, ensize_K=K, rN_seed=2103)
    t = time.perf_counter() - t0
    tab = dev.profile_table()
    print("resident again: %.1f ms" % (t * 1e3), sorted(((k, round(v[0], 2)) for k, v in tab.items()), key=lambda kv: -kv[1])[:12], flush=True)
del X, S
import gc; gc.collect()
for it in range(2):
    t0 = time.perf_counter()
    dev.SHARP_dev(dX, ensize_K=K, rN_seed=2103)
    print("resident after freeing the host matrices: %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
