"""SHARP_unlimited on ten sparse host blocks (cfg3 shape, uploads pipelined under the clustering) CALLS times: the same labels every time, seconds per call,
free device memory and thread count before / after.  usage: soak_host_blocks.py [calls=30]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
import sharp_amd
from sharp_amd import device as dev
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 30
sharp_amd.init(0)
B, nb, m = 10, 50000, 20000
blocks = []
for b in range(B):
    x = torch.empty((nb, m), dtype=torch.float32, device="cuda"); dev.synth_fill(x, 20261003, b * nb, 12, 1000)
    blocks.append(sp.csc_matrix(np.asfortranarray(x.cpu().numpy().T.astype(np.float64))))
    del x
torch.cuda.empty_cache()
ref, same, ts = None, 0, []
free0 = thr0 = None
for it in range(calls + 2):
    t0 = time.perf_counter()
    res = sharp_amd.SHARP_unlimited(blocks, ensize_K=5, rN_seed=2103, viewflag=False)
    dt = time.perf_counter() - t0
    if it < 2:
        ref = res["pred_clusters"]; free0 = torch.cuda.mem_get_info()[0]; thr0 = threading.active_count() if False else len(os.listdir("/proc/self/task"))
        continue
    ts.append(dt); same += int(np.array_equal(ref, res["pred_clusters"]))
ts = np.array(ts) * 1e3
print("%d calls: labels identical to the first call's in %d; min %.1f median %.1f max %.1f ms (%.2f M cells/s at the median); free memory %.2f -> %.2f GB; process threads %d -> %d"
      % (calls, same, ts.min(), np.median(ts), ts.max(), B * nb / np.median(ts) / 1e3, free0 / 1e9, torch.cuda.mem_get_info()[0] / 1e9, thr0, len(os.listdir("/proc/self/task"))))
