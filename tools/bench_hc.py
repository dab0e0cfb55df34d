"""Scratch: base-clustering stage times (hclust, GEMM, stats) of SHARP_dev vs number of concurrent tasks (K x 25 folds)."""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import os
import sharp_amd
from sharp_amd import _lib as _L
if os.environ.get('SHARP_VARIANT'):
    _L._SO = os.path.join(os.path.dirname(_L._SO), 'variants', 'libsharp_hip_%s.so' % os.environ['SHARP_VARIANT'])
from sharp_amd import device as dev

sharp_amd.init(0)
n, m = 50000, 20000
dX = torch.empty((n, m), dtype=torch.float32, device="cuda")
dev.synth_fill(dX, 20261003, 0, 12, 1000)
Ks = [int(a) for a in sys.argv[1:]] or [5, 10, 15, 20]
for K in Ks:
    dev.SHARP_dev(dX, ensize_K=K, rN_seed=2103)
    dev.profile(True)
    reps = 2
    for _ in range(reps):
        dev.SHARP_dev(dX, ensize_K=K, rN_seed=2103)
    prof = dev.profile_table()
    keys = ["hclust", "corr_dist_gemm", "sil_ch_stats", "cluster_sums_gemm", "row_cluster_dot_gemm", "onehot", "row_prep", "rp_stage", "host:sharp_large_total"]
    print("K=%d tasks=%d  " % (K, K * 25) + "  ".join("%s %.2f" % (k, prof.get(k, (0, 0))[0] / reps) for k in keys), flush=True)
    dev.profile(False)
