"""Sizing of the "lazy" (append-only) forms of the bulk-synchronous agglomeration on the REAL round structure of a base-clustering task (CPU,
numpy; not part of the product).  One 2000-cell fold of the bench data is projected and clustered by reciprocal-nearest-neighbour rounds exactly
as hclust_rnn_kernel does (ward.D); per round the script records the alive clusters a, the pairs P, the rows whose nearest neighbour died
(R1) and those whose cached second neighbour died too (R2), and prices four designs in matrix entries moved (units of n^2):
  full rewrite   every round reads a^2 and writes (a - P)^2                                   -- what hclust_rnn_kernel does
  lazy rounds    a round appends P new rows (read 2 rows, write 1: 3 P L), transposes them into the survivors' tails (2 P a) and rescans the
                 R rows whose neighbour died (R L); L = stored row length (alive + dead columns); every c-th round is a full rewrite
                 ("merged compaction"), or a separate compaction pass follows every c rounds
usage: python tools/sim_lazy_agglomeration.py [marker genes per cluster = 1000]          """
import sys, numpy as np
sys.path.insert(0,'/root/repo')
from oracle import pyoracle as orc
orc.build()
m, nt, p = 20000, 2000, 474
nmark = int(sys.argv[1]) if len(sys.argv)>1 else 1000
X = orc.synth_fill(20261003, m, 0, nt, 12, nmark)
E = orc.project(X, orc.ranM(m, p, 50+2103+1), True)
Ec = E - E.mean(1, keepdims=True); Ec /= np.sqrt((Ec*Ec).sum(1, keepdims=True))
D0 = 1 - Ec @ Ec.T
np.fill_diagonal(D0, np.inf)
def lw(di, dj, ni_, nj_, nk, dij_):
    return ((ni_+nk)*di + (nj_+nk)*dj - nk*dij_) / (ni_+nj_+nk)
# record per round (a, P, R_full_with_top1, R_full_with_top2)
D = D0.copy(); n = nt; size = np.ones(n); a = n
hist = []
# top-2 caching state: second[k] = index of valid 2nd NN or -1
order = np.argsort(D, axis=1)[:, :2]
nn = order[:,0].copy(); second = order[:,1].copy()
while a > 2:
    idx = np.arange(a)
    rec = (nn[nn] == idx) & (nn > idx)
    I = idx[rec]; J = nn[rec]; P = len(I)
    merged = np.zeros(a, bool); merged[I] = True; merged[J] = True
    dead_nn = merged[nn] & ~merged
    R1 = int(dead_nn.sum())
    # with top-2: full rescan needed only if second is invalid(-1) or dead
    need_full = dead_nn & ((second < 0) | merged[np.maximum(second,0)])
    R2 = int(need_full.sum())
    hist.append((a, P, R1, R2))
    keep = idx[~merged]; nb = a - P
    ni, nj = size[I], size[J]; dij = D[I, J]
    Dk = D[np.ix_(keep, keep)]
    newk = lw(D[np.ix_(I, keep)], D[np.ix_(J, keep)], ni[:,None], nj[:,None], size[keep][None,:], dij[:,None])
    t1 = lw(D[np.ix_(I, I)], D[np.ix_(J, I)], ni[:,None], nj[:,None], size[I][None,:], dij[:,None])
    t2 = lw(D[np.ix_(I, J)], D[np.ix_(J, J)], ni[:,None], nj[:,None], size[J][None,:], dij[:,None])
    nn_new = lw(t1, t2, size[I][None,:], size[J][None,:], (ni+nj)[:,None], dij[None,:])
    np.fill_diagonal(nn_new, np.inf); nn_new = np.minimum(nn_new, nn_new.T)
    Dn = np.block([[Dk, newk.T],[newk, nn_new]])
    # update top-2 for kept rows under the caching rule
    remap = -np.ones(a, int); remap[keep] = np.arange(len(keep))
    nk = len(keep)
    true_order = np.argsort(Dn, axis=1)[:, :2]
    nn2 = true_order[:,0].copy(); sec2 = -np.ones(nk, int)
    for r, k in enumerate(keep):
        if not merged[nn[k]]:
            # NN alive: unchanged; second stays valid if alive and no new cluster is closer than it
            s = second[k]
            if s >= 0 and not merged[s]:
                # candidates for second: old second vs new clusters
                newmin = newk[:, r].min() if P else np.inf
                if newmin < D[k, s]: sec2[r] = nk + int(newk[:, r].argmin())   # valid: everything old is >= old second
                else: sec2[r] = remap[s]
            else:
                sec2[r] = -1   # lost; could be recovered only by a rescan
        else:
            s = second[k]
            if s >= 0 and not merged[s]:
                # new NN among {old second, new clusters}; second valid only if both best are from this set and ordering known
                cand = np.concatenate([[D[k, s]], newk[:, r]])
                o = np.argsort(cand)[:2]
                # second valid if it is a new cluster closer than..: only if o[1]'s value <= D[k,s] (i.e. old second is among top2) or both new < old second
                if cand[o[1]] <= D[k, s]: sec2[r] = (remap[s] if o[1]==0 else nk + o[1]-1)
                else: sec2[r] = -1
            else:
                sec2[r] = true_order[r,1]   # full rescan: top-2 rebuilt
    sec_full = true_order[:,1].copy()
    sec_full[:nk] = sec2
    D = Dn; size = np.concatenate([size[keep], ni+nj]); a = nb; nn = nn2; second = sec_full
n2 = float(nt*nt)
def total(c_every, merged_compaction, top2):
    tot = 0; L = nt; since = 0
    for (a, P, R1, R2) in hist:
        R = R2 if top2 else R1
        # rows with dead nn but valid second: cost P (appendix scan) each
        cheap = (R1 - R2) if top2 else 0
        lazy = 3*P*L + 2*P*a + R*L + cheap*P
        nb = a - P
        if merged_compaction and since == c_every-1:
            tot += a*L + nb*nb; L = nb; since = 0
        else:
            tot += lazy; L += P; since += 1
            if (not merged_compaction) and since == c_every:
                tot += nb*L + nb*nb; L = nb; since = 0
    return tot/n2
print("rounds", len(hist), " full-rewrite model:", round(sum(a*a+(a-P)**2 for a,P,_,_ in hist)/n2,2))
for c in (3,4,5,6,8):
    print("c=%d: merged-compaction top1 %.2f top2 %.2f | separate-compaction top1 %.2f top2 %.2f" % (c, total(c,True,False), total(c,True,True), total(c,False,False), total(c,False,True)))
print("first rounds (a,P,R1,R2):", hist[:8])
