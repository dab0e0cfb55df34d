"""Full-size label parity against the oracle for the two regimes VERDICT r04 names (run ON THE GPU BOX; minutes of oracle time each):

  cfg4_share  one true per-GPU share of BASELINE.json configs[3]: 162 500 cells x 27 000 genes, K = 5, p = 508 -- the n >= 1e5 branch of
              SHARP_large (no reshuffle, R/SHARP.R:504-507,777-783; ~82 folds; within-block sMetaC over ~3 300 fold clusters with minN = 10,
              R/sMetaC.R:103-109) through sharp_unlimited_block_dev, vs oracle.SHARP(..., reduced_ndim = 508)
  cfg2_ch     the CH-decided data set of SURVEY.md 8d at BASELINE.json configs[1]'s size: 50 000 x 20 000, K = 15, 400 marker genes per
              planted cluster (base max median silhouette 0.26-0.30 <= sil.thre: which.max(CHind) / the height-gap rule choose k,
              R/get_opt_hclust.R:194-210) through sharp_SHARP_dev, vs oracle.SHARP
  cfg3_block_ch  one cfg3 block (50 000 x 20 000, K = 5, p = 474) of the CH-decided data set
  cfg3_block  the same block of the bench's own data set (1 000 marker genes: the silhouette rule decides)
  cfg2        BASELINE.json configs[1] whole on the bench's data set: 50 000 x 20 000, K = 15 (what tests/test_configs_gpu.py::test_cfg2_full_size_matches_oracle runs)

Every mode also compares the two decision logs (sharp_last_decisions against the oracle's) decision for decision and
prints the smallest margins per level.

usage: python tools/parity_fullsize.py cfg4_share|cfg2|cfg2_ch|cfg3_block_ch|cfg3_block [threads]      (prints a report; kept as profiles/r0N_*_parity.txt)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import sharp_amd
from oracle import pyoracle as orc
from sharp_amd import device as dev
from sharp_amd.api import ARI

SEED, RN = 20261003, 2103
what = sys.argv[1] if len(sys.argv) > 1 else "cfg2_ch"
threads = int(sys.argv[2]) if len(sys.argv) > 2 else min(len(os.sched_getaffinity(0)), 32)
sharp_amd.init(0)
orc.build()
n, m, K, p, nmark = {"cfg4_share": (162500, 27000, 5, 508, 1000), "cfg2_ch": (50000, 20000, 15, 0, 400),
                     "cfg3_block_ch": (50000, 20000, 5, 474, 400), "cfg3_block": (50000, 20000, 5, 474, 1000), "cfg2": (50000, 20000, 15, 0, 1000)}[what]
dX = torch.empty((n, m), dtype=torch.float32, device="cuda")
dev.synth_fill(dX, SEED, 0, 12, nmark)
truth = dev.synth_labels(SEED, 0, n, 12)
dev.profile(True)
sharp_amd.decision_log(True)
t0 = time.perf_counter()
if what == "cfg4_share":
    proj = sharp_amd.Projector(m, p, [50 + RN + k for k in range(1, K + 1)])
    pred, means, counts = dev.unlimited_block_dev(dX, p, proj.handle, K, RN)
    proj.close()
else:
    pred, info = dev.SHARP_dev(dX, ensize_K=K, reduced_ndim=p, rN_seed=RN)
t_gpu = time.perf_counter() - t0
got = sharp_amd.last_decisions()
sharp_amd.decision_log(False)
tab = dev.profile_table()
dev.profile(False)
rules = {k.replace("host:level_", ""): v[1] for k, v in tab.items() if k.startswith("host:level_")}
print("%s: %d cells x %d genes, K = %d, p = %s, %d marker genes per planted cluster" % (what, n, m, K, p or "from n", nmark))
print("GPU: %.3f s (first call, workspaces cold); %d clusters; ARI vs planted truth %.4f" % (t_gpu, pred.max(), ARI(truth, pred)["HA"]))
print("GPU level rules (R/get_opt_hclust.R:162-229) -- base tasks: %s; meta tasks: %s" % (
    {k[8:]: v for k, v in rules.items() if k.startswith("base_by_")}, {k[8:]: v for k, v in rules.items() if k.startswith("meta_by_")}), flush=True)
X = dX.cpu().numpy().T.astype(np.float64)                    # (genes, cells) column-major
del dX
torch.cuda.empty_cache()
print("oracle: %d threads, X = %.1f GB fp64 ..." % (threads, X.nbytes / 1e9), flush=True)
orc.stage_seconds()
t0 = time.perf_counter()
orc.decision_log(True)
ref = orc.SHARP(X, K=K, reduced_ndim=p, rN_seed=RN, nthreads=threads, want_view=False)
want = orc.last_decisions()
orc.decision_log(False)
t_or = time.perf_counter() - t0
print("oracle: %.1f s (%.0f cells/s), rc = %d, %d clusters, stages %s" % (t_or, n / t_or, ref["rc"], ref["pred_clusters"].max(), orc.stage_seconds()))
same = np.array_equal(pred, ref["pred_clusters"])
ari = ARI(ref["pred_clusters"], pred)["HA"]
print("labels identical to the oracle's, cell for cell: %s   (ARI GPU vs oracle %.6f; %d of %d cells differ)"
      % (same, ari, int((pred != ref["pred_clusters"]).sum()), n))
if what == "cfg4_share" and same:
    print("n >= 1e5 branch: no reshuffle (R/SHARP.R:504-507), %d folds, cluster sizes %s" % (-(-n // 2000), np.bincount(pred)[1:].tolist()))
# the decision logs, decision for decision
exact = [0, 1, 2, 3, 4, 5, 6, 7, 12, 13]
logs_same = got.shape == want.shape and bool((got[:, exact] == want[:, exact]).all())
print("decision logs: %d GPU rows, %d oracle rows; call, rule, chosen k, exact ties, override identical in every row: %s" % (len(got), len(want), logs_same))
if got.shape == want.shape:
    for c, name in ((8, "deciding maximum"), (9, "runner-up"), (10, "max(msil) - sil.thre")):
        ok = ~np.isnan(got[:, c]) & ~np.isnan(want[:, c])
        sil = ok & (got[:, 5] == 0)
        print("  max |GPU - oracle| of the %s: silhouette-decided %.3g, all (CH values relative) %.3g" % (
            name, np.abs(got[sil, c] - want[sil, c]).max() if sil.any() else 0.0,
            (np.abs(got[ok, c] - want[ok, c]) / np.where(got[ok, 5] >= 1, np.abs(want[ok, c]) if c < 10 else 1.0, 1.0)).max() if ok.any() else 0.0))
print("margins per level (GPU log): %s" % sharp_amd.decision_margins(got))
print("margins per level (oracle log): %s" % sharp_amd.decision_margins(want))
sys.exit(0 if same and logs_same else 1)
