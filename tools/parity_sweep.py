"""Randomised label-parity sweep GPU vs oracle (different data seeds, sizes, ensemble sizes, methods).
usage: parity_sweep.py SEED TRIALS [mixed | weak | unlimited | tpm]      (tpm: mixed sizes on TPM-like doubles -- fp64 blocks in HBM, every non-zero through the
RP kernel's value slots, round 5)"""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import sharp_amd
from oracle import pyoracle as orc

sharp_amd.init(0)
orc.build()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = 0
mixed = len(sys.argv) > 3 and sys.argv[3] in ("mixed", "weak", "tpm")
tpm = len(sys.argv) > 3 and sys.argv[3] == "tpm"      # also SHARP_small sizes and n > 1e4 (small-cluster merge)
weak = len(sys.argv) > 3 and sys.argv[3] == "weak"   # few marker genes: median silhouettes <= 0.35, the CH / height-gap branches decide
for trial in range(0 if (len(sys.argv) > 3 and sys.argv[3] == "unlimited") else (int(sys.argv[2]) if len(sys.argv) > 2 else 8)):
    seed = int(rng.integers(1, 2**31 - 1))
    lo, hi = [(300, 3000), (5200, 9000), (10001, 24000)][int(rng.integers(0, 3))] if mixed else (5200, 9000)
    n = int(rng.integers(lo, hi))
    m = int(rng.integers(1500, 3000))
    G = int(rng.integers(3, 9))
    K = int(rng.choice([3, 5, 7]))
    hm = str(rng.choice(["ward.D", "ward.D", "average", "complete", "ward.D2"]))
    rs = int(rng.integers(1, 5000))
    nmark = int(rng.integers(15, 60)) if weak else max(50, m // (2 * G))
    X = orc.synth_fill(seed, m, 0, n, G, nmark)
    if tpm:
        glen = 0.5 + np.random.default_rng(seed).random((m, 1)) * 4.0
        X = X / glen
        X = X / np.maximum(X.sum(0, keepdims=True), 1e-300) * 1e6
    t0 = time.time()
    ref = orc.SHARP(X, K=K, rN_seed=rs, hmethod=hm, nthreads=8)
    t1 = time.time()
    res = sharp_amd.SHARP(X, ensize_K=K, rN_seed=rs, hmethod=hm, forview=False, logflag=False)   # logflag=False: skip testlog (unseeded sample in the reference), log2 on, as the oracle does
    same = np.array_equal(res["pred_clusters"], ref["pred_clusters"])
    ari = orc.adjusted_rand(res["pred_clusters"], ref["pred_clusters"])["HA"]
    bad += not same
    print("trial %d seed=%d n=%d m=%d G=%d K=%d %s rN=%d%s: identical=%s ARI=%.6f clusters=%d  (oracle %.1fs)"
          % (trial, seed, n, m, G, K, hm, rs, " TPM (x_storage %d)" % sharp_amd.lib().sharp_x_storage() if tpm else "", same, ari, len(set(ref["pred_clusters"].tolist())), t1 - t0), flush=True)
if len(sys.argv) > 3 and sys.argv[3] == "unlimited":      # SHARP_unlimited on 2-4 ragged blocks (and SHARP_unlimited2 on every other trial)
    bad = 0
    for trial in range(int(sys.argv[2])):
        seed = int(rng.integers(1, 2**31 - 1))
        m, G, K = int(rng.integers(1500, 2600)), int(rng.integers(3, 9)), int(rng.choice([3, 5]))
        sizes = [int(rng.integers(5200, 9000)) for _ in range(int(rng.integers(2, 5)))]
        rs = int(rng.integers(1, 5000))
        blocks, c0 = [], 0
        for nb in sizes:
            blocks.append(orc.synth_fill(seed, m, c0, nb, G, max(50, m // (2 * G))))
            c0 += nb
        if trial % 2 == 0:
            ref = orc.SHARP_unlimited(blocks, K=K, rN_seed=rs, nthreads=8)
            res = sharp_amd.SHARP_unlimited(blocks, viewflag=False, ensize_K=K, rN_seed=rs)
            name = "SHARP_unlimited"
        else:
            ref = orc.SHARP_unlimited2(blocks, K=K, rN_seed=rs, nthreads=8)
            res = sharp_amd.SHARP_unlimited2(blocks, ensize_K=K, rN_seed=rs, forview=False, logflag=False)
            name = "SHARP_unlimited2"
        same = np.array_equal(res["pred_clusters"], ref["pred_clusters"])
        bad += not same
        print("trial %d %s seed=%d blocks=%s m=%d G=%d K=%d rN=%d: identical=%s ARI=%.6f" % (trial, name, seed, sizes, m, G, K, rs, same,
              orc.adjusted_rand(res["pred_clusters"], ref["pred_clusters"])["HA"]), flush=True)
print("mismatching runs:", bad)
