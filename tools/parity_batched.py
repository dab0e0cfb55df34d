"""SHARP_unlimited large enough for the batched base-clustering path (more than 512 base tasks) against the oracle:
8 ragged blocks of ~28 000 cells x 1500 genes, K = 5 (560 tasks).  usage: parity_batched.py [seed]"""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import sharp_amd
from oracle import pyoracle as orc

sharp_amd.init(0)
orc.build()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 20261003
B, nb, m, K, G = 8, 28000, 1500, 5, 6
blocks, c0 = [], 0
for b in range(B):
    n = nb + 37 * b
    blocks.append(orc.synth_fill(seed, m, c0, n, G, 200))
    c0 += n
t0 = time.time()
res = sharp_amd.SHARP_unlimited(blocks, ensize_K=K, rN_seed=2103)
t1 = time.time()
ref = orc.SHARP_unlimited(blocks, K=K, rN_seed=2103, nthreads=30)
t2 = time.time()
same = np.array_equal(res["pred_clusters"], ref["pred_clusters"])
print("batched SHARP_unlimited, %d cells in %d blocks: identical=%s ARI=%.6f clusters=%d  (library %.2f s incl. upload, oracle %.1f s)"
      % (c0, B, same, orc.adjusted_rand(res["pred_clusters"], ref["pred_clusters"])["HA"], len(set(ref["pred_clusters"].tolist())), t1 - t0, t2 - t1))
