"""Stage-by-stage comparison GPU vs oracle on one SHARP_large case (debugging aid)."""
import sys
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import sharp_amd as sa
from oracle import pyoracle as orc
sa.init(0); orc.build()
seed, n, m, G, K, hm, rs = 1814811279, 5808, 2786, 6, 3, "ward.D", 2224
X = orc.synth_fill(seed, m, 0, n, G, max(50, m // (2 * G)))
p = int(np.ceil(np.log2(n) / 0.04))
reind = orc.sample_perm(50, n)
folds, T = orc.make_folds(n, 2000)
Xs = X[:, reind - 1]
enrp_g, enrp_o = np.zeros((n, K), int), np.zeros((n, K), int)
for k in range(1, K + 1):
    tern = orc.ranM(m, p, 50 + rs + k)
    pr = sa.ranM2(m, p, 50 + rs + k)
    for t in range(1, T + 1):
        idx = np.nonzero(folds == t)[0]
        Eo = orc.project(Xs[:, idx], tern, True)
        Eg = pr.project(Xs[:, idx], True)
        de = np.abs(Eo - Eg).max() / np.abs(Eo).max()
        ro = orc.getrowColor(Eo, hm, height_Ntimes=2.0)
        rg = sa.getrowColor(Eo, hmethod=hm, height_Ntimes=2.0)          # same input E (the oracle's) to isolate the clustering
        rg2 = sa.getrowColor(Eg, hmethod=hm, height_Ntimes=2.0)
        ho = orc.get_opt_hclust(Eo, hm, height_Ntimes=2.0)
        hg = sa.get_opt_hclust(Eo, hmethod=hm, height_Ntimes=2.0)
        same = np.array_equal(ro["rowColor"], rg["rowColor_id"])
        same2 = np.array_equal(ro["rowColor"], rg2["rowColor_id"])
        print("k=%d t=%d nt=%d  E rel err %.2e  labels(same E) %s  labels(own E) %s  optN %d/%d  maxsil %.6f/%.6f branch %d/%d msil maxdiff %.2e CH reldiff %.2e"
              % (k, t, idx.size, de, same, same2, ho["optN"], hg["optN_cluster"], ho["maxsil"], hg["maxsil"], ho["branch"], hg["branch"],
                 np.abs(ho["msil"] - hg["msil"]).max(), np.abs((ho["CHind"] - hg["CHind"]) / ho["CHind"]).max()), flush=True)
        enrp_o[idx, k - 1] = ro["rowColor"]; enrp_g[idx, k - 1] = rg2["rowColor_id"]
# ---- downstream stages on the oracle's base labels (identical above) ----
maxN = max(40, -(-n // 5000))
E1 = np.zeros((n, p))
for k in range(1, K + 1):
    tern = orc.ranM(m, p, 50 + rs + k)
    for t in range(1, T + 1):
        idx = np.nonzero(folds == t)[0]
        E1[idx] += orc.project(Xs[:, idx], tern, True).reshape(idx.size, p) if False else np.asarray(orc.project(Xs[:, idx], tern, True)).reshape(idx.size, -1)
E1 /= K
fc_o, fc_g = np.zeros(n, int), np.zeros(n, int)
for t in range(1, T + 1):
    idx = np.nonzero(folds == t)[0]
    wo = orc.wMetaC(enrp_o[idx], hm, 0, 2, maxN, 0.35, 2.0)
    wg = sa.wMetaC(enrp_o[idx], hmethod=hm, enN_cluster=0, minN_cluster=2, maxN_cluster=maxN, sil_thre=0.35, height_Ntimes=2.0, debug=True)
    print("fold %d wMetaC same %s  ncl %d/%d  allC %d" % (t, np.array_equal(wo["finalC"], wg["finalC"]), wo["finalC"].max(), np.max(wg["finalC"]), wo["allC"]),
          " w1 maxdiff %.2e  S maxdiff %.2e" % (np.abs(wo["w1"] - wg["w1"]).max() if "w1" in wg else -1, np.abs(wo["S"] - wg["S"]).max() if "S" in wg else -1), flush=True)
    fc_o[idx] = t * 65536 + wo["finalC"]; fc_g[idx] = t * 65536 + np.asarray(wg["finalC"])
so = orc.sMetaC(fc_o, E1, hm, 0, 2, maxN, 0.35, 2.0)
sg = sa.sMetaC(fc_o, E1, hmethod=hm, finalN_cluster=0, minN_cluster=2, maxN_cluster=maxN, sil_thre=0.35, height_Ntimes=2.0)
print("sMetaC same %s  nC %d  ncl %d/%d" % (np.array_equal(so["finalColor"], sg["finalColor"]), so["nC"], so["finalColor"].max(), sg["finalColor"].max()))
print("tf o", so["tf"]); print("tf g", sg["tf"])
