"""Generates tests/golden/oracle_small.npz from the CPU oracle (oracle/sharp_oracle.c).

The reference (R) cannot run in the build container and ships no vectors, so these are regression vectors of
the oracle itself plus the externally known R outputs it was pinned on (set.seed()/runif()/sample(), the familiar
console values).  Re-run: python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as orc  # noqa: E402

SEED = 20261003
out = {}
# externally known R console outputs (not produced by the oracle)
out["R_runif_seed1"] = np.array([0.2655087, 0.3721239, 0.5728534])
out["R_runif_seed42"] = np.array([0.9148060, 0.9370754])
out["R_sample10_seed42"] = np.array([1, 5, 10, 8, 2, 4, 6, 9, 7, 3])
out["R_sample10_seed123"] = np.array([3, 10, 2, 8, 6, 9, 1, 7, 5, 4])
# oracle regression vectors
m, n, G, nm = 600, 80, 3, 150
X = orc.synth_fill(SEED, m, 0, n, G, nm)
p = int(np.ceil(np.log2(n) / 0.04))
t = orc.ranM(m, p, 2154)
gi, ci = np.nonzero(t)
out["synth_X"] = X.astype(np.float32)
out["ranM_gene"], out["ranM_col"], out["ranM_sign"] = gi.astype(np.int32), ci.astype(np.int32), t[gi, ci]
E = orc.project(X, t, True)
out["E"] = E
r = orc.get_opt_hclust(E)
out["hc_height"], out["hc_msil"], out["hc_CH"], out["hc_f"], out["hc_v"] = r["height"], r["msil"], r["CHind"], r["f"], r["v"]
s = orc.SHARP_small(X, K=3, rN_seed=2103)
out["small_pred"], out["small_enrp"], out["small_x0"] = s["pred_clusters"], s["enrp"], s["x0"]
w = orc.wMetaC(s["enrp"])
out["wm_w1"], out["wm_S"], out["wm_finalC"] = w["w1"], w["S"], w["finalC"]
big = orc.synth_fill(SEED, m, 0, 260, G, nm)
L = orc.SHARP(big, K=3, base_ncells=100, partition_ncells=80, rN_seed=2103)
out["large_X_cells"] = np.array([260])
out["large_pred"] = L["pred_clusters"]
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_small.npz"), **out)
print("wrote", {k: v.shape for k, v in out.items()})
