"""Times get_marker_genes' per-gene pass on the benchmark block (50 000 cells x 20 000 genes, planted labels)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import sharp_amd
from sharp_amd import device as dev

sharp_amd.init(0)
n, m, G = 50000, 20000, 12
dX = torch.empty((n, m), dtype=torch.float32, device="cuda")
dev.synth_fill(dX, 20261003, 0, G, 1000)
lab = dev.synth_labels(20261003, 0, n, G) + 1
torch.cuda.synchronize()
for rep in range(3):
    dev.profile(True)
    t0 = time.perf_counter()
    out = dev.marker_genes_dev(dX, lab, G)
    dt = time.perf_counter() - t0
    tab = dev.profile_table()
    print("run %d: %.1f ms  %s  genes with auc > 0.7: %d" % (rep, dt * 1e3, {k: round(v[0], 2) for k, v in tab.items() if k.startswith("marker")},
                                                          int((out[:, 0] > 0.7).sum())), flush=True)
