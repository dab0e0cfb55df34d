"""Three SHARP_unlimited calls on ten sparse host blocks (cfg3 shape), for a kernel trace:
   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/host_tr -- python3 tools/host_trace.py ; tools/timeline.py gpurun_out/host_tr 300 -1"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
import sharp_amd
from sharp_amd import device as dev
sharp_amd.init(0)
B, nb, m = 10, 50000, 20000
blocks = []
for b in range(B):
    x = torch.empty((nb, m), dtype=torch.float32, device="cuda"); dev.synth_fill(x, 20261003, b * nb, 12, 1000)
    blocks.append(sp.csc_matrix(np.asfortranarray(x.cpu().numpy().T.astype(np.float64))))
    del x
import time
for it in range(3):
    t0 = time.perf_counter()
    sharp_amd.SHARP_unlimited(blocks, ensize_K=5, rN_seed=2103, viewflag=False)
    print("call %d: %.3f s" % (it, time.perf_counter() - t0), flush=True)
