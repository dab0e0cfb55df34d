"""Host-side mirror of the reference's R interface for the hot path.

Function names, argument names/meaning and error behaviour follow the R package
(dots in R argument names become underscores).  Everything is computed by
libsharp_hip.so through ctypes; arrays are numpy on the host boundary."""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import SharpError, check, lib

__all__ = ["ranM", "ranM2", "RPmat", "Projector", "SharpError", "get_opt_hclust", "getrowColor", "colorL", "HMETHODS",
           "wMetaC", "sMetaC", "SHARP", "SHARP_small", "SHARP_large", "SHARP_unlimited", "SHARP_unlimited2", "SHARP_unlimited3", "run_Mtimes_SHARP", "get_marker_genes", "get_marker_genes_unlimited",
           "get_marker_genes_unlimited2", "testlog", "ARI", "decision_log", "last_decisions", "decision_margins", "DECISION_FIELDS"]


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int))


class Projector:
    """K sparse ternary projectors (the `rM` list of R/SHARP.R:539-549) resident on the GPU."""

    def __init__(self, m, p, seeds):
        _lib.ensure_init()
        seeds = np.ascontiguousarray(np.atleast_1d(seeds), np.float64)
        h = C.c_int()
        check(lib().sharp_projector_create(int(m), int(p), int(seeds.size), _dp(seeds), C.byref(h)))
        self.handle = h.value
        self.m, self.p, self.K = int(m), int(p), int(seeds.size)
        self.value = float(np.sqrt(np.sqrt(m)))

    def close(self):
        if getattr(self, "handle", None):
            lib().sharp_projector_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def nnz(self):
        nn = C.c_longlong()
        check(lib().sharp_projector_info(self.handle, None, None, None, C.byref(nn)))
        return nn.value

    def triplets(self, k=0):
        """(gene, col, sign) of projector k, row-major order -- what Matrix(x0, byrow=TRUE, sparse=TRUE) stores."""
        nn = C.c_longlong()
        check(lib().sharp_projector_triplets(self.handle, int(k), None, None, None, C.byref(nn)))
        g = np.empty(nn.value, np.int32)
        c = np.empty(nn.value, np.int32)
        s = np.empty(nn.value, np.int8)
        check(lib().sharp_projector_triplets(self.handle, int(k), _ip(g), _ip(c), s.ctypes.data_as(C.POINTER(C.c_byte)),
                                             C.byref(nn)))
        return g, c, s

    def dense(self, k=0):
        """m x p float64 matrix with entries {+sqrt(s), 0, -sqrt(s)} (small sizes only)."""
        g, c, s = self.triplets(k)
        R = np.zeros((self.m, self.p))
        R[g, c] = s * self.value
        return R

    def project(self, X, logflag=True):
        """E (n, K*p): component k*p+c = (1/sqrt(p)) * t(R_k) %*% log2(X+1), X genes x cells."""
        X = np.asfortranarray(X, dtype=np.float64)
        m, n = X.shape
        E = np.empty((n, self.K * self.p), np.float64)
        check(lib().sharp_project(self.handle, _dp(X), m, n, C.c_longlong(m), int(bool(logflag)), _dp(E)))
        return E


def _seed_check(seedn):
    if not isinstance(seedn, (int, float, np.integer, np.floating)):
        raise SharpError("The seed should be a numeric!")


def ranM(scdata, p, seedn):
    """R/ranM.R:11-33 -- sparse ternary m x p projector for `scdata` (genes x cells)."""
    _seed_check(seedn)
    return Projector(np.shape(scdata)[0], p, [seedn])


def ranM2(m, p, seedn):
    """R/ranM2.R:11-35 -- as ranM but takes the number of features."""
    _seed_check(seedn)
    return Projector(m, p, [seedn])


def RPmat(scdata, p, seedn):
    """R/RPmat.R:14-47 -- list(R = projector, projmat = 1/sqrt(p) * t(R) %*% scdata) (p x n)."""
    pr = ranM(scdata, p, seedn)
    E = pr.project(scdata, logflag=False)
    return {"R": pr, "projmat": E.T.copy()}


HMETHODS = {"ward.D": 1, "single": 2, "complete": 3, "average": 4, "mcquitty": 5, "median": 6, "centroid": 7,
            "ward.D2": 8}

# R/getrowColor.R:52-58
colorL = ["red", "purple", "blue", "yellow", "green", "orange", "brown", "gray", "black", "coral", "beige", "cyan",
          "turquoise", "pink", "khaki", "magenta", "violet", "salmon", "goldenrod", "orchid", "seagreen", "slategray",
          "darkred", "darkblue", "darkcyan", "darkgreen", "darkgray", "darkkhaki", "darkorange", "darkmagenta",
          "darkviolet", "darkturquoise", "darksalmon", "darkgoldenrod", "darkorchid", "darkseagreen", "darkslategray",
          "deeppink", "lightcoral", "lightcyan"]


def _hmethod(h, flashmark=False):
    """hclust method code.  flashmark (R/get_opt_hclust.R:76-83): TRUE takes flashClust(d, "ward"), the ward.D criterion -- the same
    merges as hclust(d, "ward.D") --, and the guard `hmethod == "ward.D" || "ward.D2"` is an R error for every other method
    (`||` on a character string), reproduced here."""
    if h is None:
        h = "ward.D"
    if h not in HMETHODS:
        raise SharpError(f"invalid clustering method '{h}'")
    if flashmark:
        if h != "ward.D":
            raise SharpError("invalid 'y' type in 'x || y'")
        return 1
    return HMETHODS[h]


def get_opt_hclust(mat, hmethod=None, N_cluster=None, minN_cluster=None, maxN_cluster=None, sil_thre=None,
                   height_Ntimes=None, flashmark=False):
    """R/get_opt_hclust.R:33-244.  `mat`: (n, p) feature rows or an (n, n) symmetric similarity.
    Returns the `hres` list as a dict: f, v, maxsil, msil, CHind, height, optN_cluster (+ branch, warn)."""
    _lib.ensure_init()
    a = np.ascontiguousarray(mat, np.float64)
    if a.ndim != 2:
        raise SharpError("mat must be a matrix")
    n, p = a.shape
    minN = 2 if minN_cluster is None else int(minN_cluster)
    maxN = 40 if maxN_cluster is None else int(maxN_cluster)
    sil = 0.35 if sil_thre is None else float(sil_thre)
    hN = 2.0 if height_Ntimes is None else float(height_Ntimes)
    Ncl = 0
    if N_cluster is not None:
        if not isinstance(N_cluster, (int, float, np.integer, np.floating)):
            raise SharpError("The given N.cluster is not a numeric!")
        if N_cluster % 1 != 0:
            raise SharpError("The given N.cluster is not an integer!")
        if N_cluster < 2:
            raise SharpError("The given N.cluster is less than 2, which is not suitable for clustering!")
        Ncl = int(N_cluster)
    nk = 1 if Ncl else max(1, min(maxN, n - 1) - minN + 1)
    f = np.zeros(n, np.int32)
    v = np.zeros(n * nk, np.int32)
    msil = np.zeros(nk)
    ch = np.zeros(nk)
    height = np.zeros(max(n - 1, 1))
    maxsil = C.c_double()
    optN = C.c_int()
    nko = C.c_int()
    br = C.c_int()
    rc = check(lib().sharp_get_opt_hclust(_dp(a), n, p, _hmethod(hmethod, flashmark), Ncl, minN, maxN, C.c_double(sil),
                                          C.c_double(hN), _ip(f), _ip(v), _dp(msil), _dp(ch), C.byref(maxsil),
                                          _dp(height), C.byref(optN), C.byref(nko), C.byref(br)), allow=16)
    k = nko.value
    return {"f": f, "v": v[: n * k].reshape(k, n).T.copy(), "maxsil": maxsil.value, "msil": msil[:k], "CHind": ch[:k],
            "height": height[: n - 1], "optN_cluster": optN.value, "branch": br.value, "warn": rc}


def getrowColor(Emat, hmethod=None, indN_cluster=None, minN_cluster=2, maxN_cluster=40, sil_thre=0.35,
                height_Ntimes=None, flashmark=False):
    """R/getrowColor.R:17-121 -> dict(rowColor = colour names, maxsil, mat = Emat (unscaled))."""
    _lib.ensure_init()
    a = np.ascontiguousarray(Emat, np.float64)
    n, p = a.shape
    rc_ = np.zeros(n, np.int32)
    maxsil = C.c_double()
    check(lib().sharp_getrowColor(_dp(a), n, p, _hmethod(hmethod, flashmark), int(indN_cluster or 0), int(minN_cluster),
                                  int(maxN_cluster), C.c_double(sil_thre),
                                  C.c_double(1.0 if height_Ntimes is None else height_Ntimes), _ip(rc_), C.byref(maxsil)),
          allow=16)
    return {"rowColor": [colorL[j - 1] for j in rc_], "rowColor_id": rc_, "maxsil": maxsil.value, "mat": Emat}


def _labels_to_int(col):
    """R label strings / arbitrary hashables -> ints with equal <=> same label."""
    col = np.asarray(col)
    if col.dtype.kind in "iu":
        return col.astype(np.int32)
    _, inv = np.unique(col, return_inverse=True)
    return inv.astype(np.int32)


def wMetaC(nC, hmethod=None, enN_cluster=None, minN_cluster=None, maxN_cluster=None, sil_thre=None,
           height_Ntimes=None, debug=False):
    """R/wMetaC.R:15-226.  nC: (N, C) labels (ints or strings).  Returns dict(finalC, x0)."""
    _lib.ensure_init()
    nC = np.asarray(nC)
    if nC.ndim != 2:
        raise SharpError("nC must be an N x C matrix of cluster labels")
    N, Cc = nC.shape
    lab = np.asfortranarray(np.stack([_labels_to_int(nC[:, c]) for c in range(Cc)], 1), dtype=np.int32)
    minN = 2 if minN_cluster is None else int(minN_cluster)
    maxN = 40 if maxN_cluster is None else int(maxN_cluster)
    finalC = np.zeros(N, np.int32)
    x0 = np.zeros(N * (maxN + 2))
    ncl = C.c_int()
    allC = C.c_int()
    w1 = np.zeros(N) if debug else None
    cap = N * Cc if N * Cc < 4096 else 4096
    S = np.zeros(cap * cap) if debug else None
    tf = np.zeros(cap, np.int32) if debug else None
    rc = check(lib().sharp_wMetaC(_ip(lab), N, Cc, _hmethod(hmethod), int(enN_cluster or 0), minN, maxN,
                                  C.c_double(0.0 if sil_thre is None else sil_thre),   # R/wMetaC.R:94-97
                                  C.c_double(2.0 if height_Ntimes is None else height_Ntimes), _ip(finalC), _dp(x0),
                                  C.byref(ncl), _dp(w1), _dp(S), C.byref(allC), _ip(tf)), allow=48)
    out = {"finalC": finalC, "x0": x0[: N * ncl.value].reshape(ncl.value, N).T.copy(), "warn": rc}
    if debug:
        A = allC.value
        out.update(w1=w1, S=S[: A * A].reshape(A, A).copy(), tf=tf[:A].copy(), allC=A)
    return out


def sMetaC(rerowColor, sE1, folds=None, hmethod=None, finalN_cluster=None, minN_cluster=2, maxN_cluster=40,
           sil_thre=0.35, height_Ntimes=2.0):
    """R/sMetaC.R:17-210.  Returns dict(finalColor, tf).  `folds` is accepted and unused, as in the reference."""
    _lib.ensure_init()
    lab = np.ascontiguousarray(_labels_to_int(rerowColor))
    E = np.ascontiguousarray(sE1, np.float64)
    n, p = E.shape
    fin = np.zeros(n, np.int32)
    tf = np.zeros(n, np.int32)
    nC = C.c_int()
    rc = check(lib().sharp_sMetaC(_ip(lab), _dp(E), C.c_longlong(n), p, _hmethod(hmethod), int(finalN_cluster or 0),
                                  int(minN_cluster), int(maxN_cluster), C.c_double(sil_thre), C.c_double(height_Ntimes),
                                  _ip(fin), _ip(tf), C.byref(nC)), allow=16)
    return {"finalColor": fin, "tf": tf[: nC.value].copy(), "warn": rc}


def _is_sparse(x):
    return hasattr(x, "tocsc") and hasattr(x, "nnz")


def _csc_int_slots(blocks):
    """The @p / @i slots of a list of CSC blocks as the int32 vectors the C ABI takes (a dgCMatrix holds ints too).  scipy switches to
    int64 index arrays beyond 2^31 - 1 stored entries; such a block cannot be passed at all and must fail loudly, not wrap."""
    for q, b in enumerate(blocks):
        if int(b.indptr[-1]) >= 2**31 or b.shape[0] >= 2**31 or b.shape[1] >= 2**31 - 1:
            raise SharpError("sparse block %d: more than 2^31 - 1 stored entries, rows or columns (the C ABI, like R's dgCMatrix, "
                             "indexes with int): split the block" % (q + 1))
    return ([np.ascontiguousarray(b.indptr, np.int32) for b in blocks], [np.ascontiguousarray(b.indices, np.int32) for b in blocks])


def testlog(scExp, ncells, p, sncells=100, n_cores=None, cells=None):
    """R/SHARP.R:877-924.  The reference draws the test cells with the unseeded global RNG (:884), so its
    result is not reproducible; pass `cells` (0-based indices) to fix them."""
    sncells = min(sncells, ncells)
    if cells is None:
        cells = np.random.default_rng().permutation(ncells)[:sncells]
    if _is_sparse(scExp):
        sE = np.asfortranarray(scExp[:, np.asarray(cells)].toarray(), dtype=np.float64)
    else:
        sE = np.asarray(scExp, np.float64)[:, np.asarray(cells)]
    pr = ranM2(scExp.shape[0], p, 5)
    msil = []
    for k in (1, 2):
        E1 = pr.project(sE, logflag=(k == 2))
        msil.append(getrowColor(E1, "ward.D", None, 2, 40, 0.0, 2.0)["maxsil"])
    return bool(msil[0] < 0.75 and msil[0] >= 0.95 * msil[1])


def _enresults(pred, x0, viE, ncells, ngenes, p, K, t0, paras, forview, key="N.pred_cluster", allrpinfo=None):
    import time as _t

    uy = np.unique(pred)
    out = {"pred_clusters": pred, "unique_pred_clusters": uy, "distr_pred_clusters": {int(u): int((pred == u).sum()) for u in uy},
           key: int(uy.size)}
    if forview:
        if allrpinfo is not None:                                             # SHARP_small only (R/SHARP.R:445-449)
            out["allrpinfo"] = allrpinfo
        out["x0"] = x0
        out["viE"] = viE
    out.update({"N.cells": ncells, "N.genes": ngenes, "reduced.dim": p, "ensize.K": K,
                "time": (_t.time() - t0) / 60.0, "paras": paras})
    return out


def _allrpinfo():
    """allrpinfo of the SHARP_small run that just finished (R/SHARP.R:350-387,446): per random projection its tag "_RP<p>_<k>",
    the rowColor of every cell (colour names), N.cluster and indE = the projected cells x p matrix."""
    n, K, p = C.c_int(), C.c_int(), C.c_int()
    check(lib().sharp_last_rpinfo(C.byref(n), C.byref(K), C.byref(p), None, None))
    enrp = np.zeros((K.value, n.value), np.int32)
    indE = np.zeros((n.value, K.value * p.value))
    check(lib().sharp_last_rpinfo(None, None, None, _ip(enrp), _dp(indE)))
    out = []
    for k in range(K.value):
        rc_ = [colorL[j - 1] for j in enrp[k]]
        out.append({"tag": "_RP%d_%d" % (p.value, k + 1), "rowColor": rc_, "N.cluster": len(set(rc_)),
                    "indE": indE[:, k * p.value:(k + 1) * p.value].copy()})
    return out


def _run_sharp(X, K, p, base_ncells, partition_ncells, hmethod, N_cluster, enpN, indN, minN, maxN, sil_thre,
               height_Ntimes, flag, rM, rN_seed, forview, flashmark=False):
    sparse = _is_sparse(X)
    if sparse:                                                            # dgCMatrix-style input: only the non-zeros are uploaded
        X = X.tocsc()
        if X.nnz >= 2**31:
            raise SharpError("sparse input: more than 2^31 - 1 stored entries (the limit of a dgCMatrix as well)")
        cp = np.ascontiguousarray(X.indptr, np.int32)
        ri = np.ascontiguousarray(X.indices, np.int32)
        xv = np.ascontiguousarray(X.data, np.float64)
    else:
        X = np.asfortranarray(X, dtype=np.float64)
    m, n = X.shape
    pred = np.zeros(n, np.int32)
    p_eff = p if p else int(np.ceil(np.log2(n) / 0.04))
    viE = np.zeros((n, p_eff)) if forview else None
    capc = max(int(maxN or 0), 40, (n + 4999) // 5000) + 2
    x0 = np.zeros(n * capc) if forview else None
    npred, x0c, pu, Ku, path = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
    entry = ((lib().sharp_SHARP_csc, (_ip(cp), _ip(ri), _dp(xv), m, C.c_longlong(n))) if sparse else
             (lib().sharp_SHARP, (_dp(X), m, C.c_longlong(n), C.c_longlong(m))))
    rc = check(entry[0](*entry[1], int(K or 0), int(p or 0),
                                 int(base_ncells or 0), int(partition_ncells or 0), _hmethod(hmethod, flashmark), int(N_cluster or 0),
                                 int(enpN or 0), int(indN or 0), int(minN or 0), int(maxN or 0),
                                 C.c_double(-1.0 if sil_thre is None else sil_thre),
                                 C.c_double(0.0 if height_Ntimes is None else height_Ntimes), int(bool(flag)),
                                 int(rM.handle if isinstance(rM, Projector) else 0), C.c_double(rN_seed), _ip(pred),
                                 C.byref(npred), _dp(viE), _dp(x0), capc, C.byref(x0c), C.byref(pu), C.byref(Ku),
                                 C.byref(path)), allow=48)
    x0m = x0[: n * x0c.value].reshape(x0c.value, n).T.copy() if forview else None
    info = _allrpinfo() if forview and path.value == 0 else None          # SHARP_small only (R/SHARP.R:446)
    return pred, x0m, viE, pu.value, Ku.value, path.value, rc, info


def SHARP(scExp, exp_type=None, ensize_K=None, reduced_ndim=None, base_ncells=None, partition_ncells=None, hmethod=None,
          N_cluster=None, enpN_cluster=None, indN_cluster=None, minN_cluster=None, maxN_cluster=None, sil_thre=None,
          height_Ntimes=None, flashmark=False, logflag=None, sncells=None, n_cores=None, forview=True, prep=None,
          rM=None, rN_seed=None, gene_names=None, cell_names=None, testlog_cells=None):
    """R/SHARP.R:44-318.  scExp: (genes, cells).  Returns the `enresults` list as a dict."""
    import time as _t
    import warnings

    t0 = _t.time()
    if scExp is None:
        raise SharpError("No expression data is provided!")
    _lib.ensure_init()
    sparse = _is_sparse(scExp)
    if sparse:                                                            # scipy.sparse stands in for the Matrix package's dgCMatrix
        X = scExp.tocsc().astype(np.float64)
        if not X.has_canonical_format:
            X = X.copy()
            X.sum_duplicates()
    else:
        X = np.asarray(scExp, dtype=np.float64)                           # never modified in place (8 GB at cfg2)
    ngenes, ncells = X.shape
    if prep is None:
        prep = ncells < 1e4                                               # :74-80
    if gene_names is not None:                                            # :83-88
        _, first = np.unique(np.asarray(gene_names), return_index=True)
        if first.size < len(gene_names):
            warnings.warn(f"{len(gene_names) - first.size} duplicated genes are found and then are removed!")
            X = X[np.sort(first)]
    if prep and sparse:                                                   # :99-106 on the stored entries
        if (X.data < 0).any():
            warnings.warn("Your expression matrix contain negative values! SHARP will replace negative values with 0!")
            X = X.copy()
            X.data[X.data < 0] = 0.0
        X = X[np.asarray(X.sum(1)).ravel() != 0]
    elif prep:
        if (X < 0).any():
            warnings.warn("Your expression matrix contain negative values! SHARP will replace negative values with 0!")
            X = np.where(X < 0, 0.0, X)
        X = X[X.sum(1) != 0]
    if exp_type is not None and exp_type not in ("CPM", "TPM"):           # :110-114
        if sparse:
            X = X.tocsc(copy=True)
            colsum = np.asarray(X.sum(0)).ravel()
            X.data = X.data / np.repeat(colsum, np.diff(X.indptr)) * 1e6
        else:
            X = X / X.sum(0, keepdims=True) * 1e6
    if rN_seed is not None:                                               # :169-179
        if not isinstance(rN_seed, (int, float, np.integer, np.floating)):
            raise SharpError("The rN.seed should be a numeric!")
        if rN_seed % 1 != 0 and rN_seed != 0.5:
            raise SharpError("The rN.seed should be an integer!")
    else:
        rN_seed = 0.5
    p = int(reduced_ndim) if reduced_ndim else int(np.ceil(np.log2(ncells) / 0.04))   # :119-122
    if logflag is None:
        logflag = ncells < 1e4                                            # :202-209
    if logflag:
        flag = testlog(X, ncells, p, 100 if sncells is None else sncells, n_cores, testlog_cells)   # :211-224
    else:
        flag = True                                                       # :225-228
    pred, x0, viE, pu, Ku, path, rc, rpinfo = _run_sharp(X, ensize_K, p, base_ncells, partition_ncells, hmethod, N_cluster,
                                                         enpN_cluster, indN_cluster, minN_cluster, maxN_cluster, sil_thre,
                                                         height_Ntimes, flag, rM, rN_seed, forview, flashmark)
    paras = {"ensize.K": Ku, "reduced.ndim": pu, "base.ncells": base_ncells or 5000,
             "partition.ncells": partition_ncells or 2000, "logmark": flag, "hmethod": hmethod or "ward.D",
             "N.cluster": N_cluster, "minN.cluster": minN_cluster or 2,
             "maxN.cluster": maxN_cluster or max(40, -(-ncells // 5000)), "sil.thre": 0.35 if sil_thre is None else sil_thre,
             "height.Ntimes": height_Ntimes or 2, "n.cores": n_cores}
    out = _enresults(pred, x0, viE, ncells, ngenes, pu, Ku, t0, paras, forview, allrpinfo=rpinfo)
    out["warn"] = rc
    out["path"] = "SHARP_large" if path else "SHARP_small"
    return out


def SHARP_small(scExp, ncells=None, ensize_K=15, reduced_ndim=None, hmethod="ward.D", N_cluster=None, indN_cluster=None,
                minN_cluster=2, maxN_cluster=40, sil_thre=0.35, height_Ntimes=2, flashmark=False, flag=True, n_cores=None,
                forview=True, rN_seed=0.5):
    """R/SHARP.R:339-454 (same argument order)."""
    import time as _t

    t0 = _t.time()
    _lib.ensure_init()
    n = np.shape(scExp)[1]
    pred, x0, viE, pu, Ku, _, rc, rpinfo = _run_sharp(scExp, ensize_K, reduced_ndim, n + 1, None, hmethod, N_cluster, None,
                                                      indN_cluster, minN_cluster, maxN_cluster, sil_thre, height_Ntimes, flag, None,
                                                      rN_seed, forview, flashmark)
    return _enresults(pred, x0, viE, n, np.shape(scExp)[0], pu, Ku, t0, {}, forview, allrpinfo=rpinfo)


def SHARP_large(scExp, ncells=None, ensize_K=5, reduced_dim=None, partition_ncells=2000, hmethod="ward.D", N_cluster=None,
                enpN_cluster=None, indN_cluster=None, minN_cluster=2, maxN_cluster=40, sil_thre=0.35, height_Ntimes=2,
                flashmark=False, flag=True, n_cores=None, forview=True, rM=None, rN_seed=0.5):
    """R/SHARP.R:478-851 (same argument order)."""
    import time as _t

    t0 = _t.time()
    _lib.ensure_init()
    n = np.shape(scExp)[1]
    pred, x0, viE, pu, Ku, _, rc, _info = _run_sharp(scExp, ensize_K, reduced_dim, 1, partition_ncells, hmethod, N_cluster,
                                                     enpN_cluster, indN_cluster, minN_cluster, maxN_cluster, sil_thre, height_Ntimes,
                                                     flag, rM, rN_seed, forview, flashmark)
    return _enresults(pred, x0, viE, n, np.shape(scExp)[0], pu, Ku, t0, {}, forview)


def SHARP_unlimited(scExp, viewflag=True, n_cores=None, ensize_K=None, N_cluster=None, minN_cluster=None,
                    maxN_cluster=None, rN_seed=None, devices=None):
    """R/SHARP_unlimited.R:29-242.  scExp: list of (genes, cells) blocks sharing the gene axis.
    devices (no reference counterpart: the reference's block loop is serial, :125-163): GPU indices; block b runs on devices[b mod N],
    one host thread and one device context per GPU inside this process (sharp_SHARP_unlimited_multi), same labels as on one GPU."""
    import time as _t
    import warnings

    t0 = _t.time()
    if scExp is None:
        raise SharpError("No expression data is provided!")
    if not isinstance(scExp, (list, tuple)):
        if isinstance(scExp, np.ndarray):                                 # :39-45
            warnings.warn("SHARP is used instead of SHARP_unlimited because the input is a matrix!")
            return SHARP(scExp)
        raise SharpError("The input should be a LIST of partitioned scRNA-seq expression matrices!")
    if len(scExp) == 1:                                                   # :47-51
        warnings.warn("SHARP is used instead of SHARP_unlimited because the length of the input is 1!")
        return SHARP(scExp[0])
    if rN_seed is not None:
        if not isinstance(rN_seed, (int, float, np.integer, np.floating)):
            raise SharpError("The rN.seed should be a numeric!")
        if rN_seed % 1 != 0:
            raise SharpError("The rN.seed should be an integer!")
    else:
        rN_seed = 0.5
    _lib.ensure_init()
    sparse = all(_is_sparse(b) for b in scExp)                            # a list of dgCMatrix-like blocks (:125-135 hands them on as they are)
    if sparse:
        blocks = [b.tocsc() for b in scExp]
        for b in blocks:
            b.sum_duplicates()
    else:
        blocks = [np.asfortranarray(b.toarray() if _is_sparse(b) else b, dtype=np.float64) for b in scExp]
    m = blocks[0].shape[0]
    ncb = np.array([b.shape[1] for b in blocks], np.int64)
    n = int(ncb.sum())
    pred = np.zeros(n, np.int32)
    npred, pu = C.c_int(), C.c_int()
    p = int(np.ceil(np.log2(n) / 0.04))                                   # :65-66, from the TOTAL number of cells
    # :215-232: above 1e5 cells enresults$viE is E1 reduced to 50 columns by one more sparse projection; the library takes that product per
    # block on the GPU (sharp_unlimited_view_dim), so that ncells x 50 doubles come back instead of ncells x p
    kdim = 50 if (viewflag and n > 1e5) else 0
    viE = np.zeros((n, kdim if kdim else p)) if viewflag else None

    def call(entry, *args):
        # the arm is one-shot and thread-local in the library, taken by the call that follows on this thread; it is set here, after all the
        # marshalling that can raise, and withdrawn if the call itself is never made
        try:
            if kdim:
                check(lib().sharp_unlimited_view_dim(kdim))
            return check(entry(*args), allow=48)
        finally:
            if kdim:
                lib().sharp_unlimited_view_dim(0)

    if sparse:
        # only the non-zeros of a block cross PCIe, block b + W while block b is clustered (sharp_SHARP_unlimited_csc_multi)
        cps, ris = _csc_int_slots(blocks)
        vxs = [np.ascontiguousarray(b.data, np.float64) for b in blocks]
        B = len(blocks)
        cpp = (C.POINTER(C.c_int) * B)(*[_ip(a) for a in cps])
        rip = (C.POINTER(C.c_int) * B)(*[_ip(a) if a.size else C.cast(None, C.POINTER(C.c_int)) for a in ris])
        vxp = (C.POINTER(C.c_double) * B)(*[_dp(a) if a.size else C.cast(None, C.POINTER(C.c_double)) for a in vxs])
        dv = np.ascontiguousarray(devices if devices is not None else [], np.int32)
        call(lib().sharp_SHARP_unlimited_csc_multi, cpp, rip, vxp, ncb.ctypes.data_as(C.POINTER(C.c_longlong)), B, m, int(ensize_K or 0),
             int(N_cluster or 0), int(minN_cluster or 0), int(maxN_cluster or 0), C.c_double(rN_seed),
             _ip(dv) if dv.size else None, int(dv.size), _ip(pred), C.byref(npred), C.byref(pu), _dp(viE))
        K = int(ensize_K or 5)
        out = _enresults(pred, None, None, n, m, pu.value, K, t0, {}, False, key="N.pred_clusters")
        if viewflag:                                                      # :215-232
            out["viE"] = viE
            out["x0"] = _one_hot(pred, npred.value)
        return out
    ptrs = (C.POINTER(C.c_double) * len(blocks))(*[_dp(b) for b in blocks])
    if devices is not None and len(devices) >= 1:
        dv = np.ascontiguousarray(devices, np.int32)
        call(lib().sharp_SHARP_unlimited_multi, ptrs, ncb.ctypes.data_as(C.POINTER(C.c_longlong)), len(blocks), m, int(ensize_K or 0),
             int(N_cluster or 0), int(minN_cluster or 0), int(maxN_cluster or 0),
             C.c_double(rN_seed), _ip(dv), len(dv), _ip(pred), C.byref(npred), C.byref(pu), _dp(viE))
    else:
        call(lib().sharp_SHARP_unlimited_view, ptrs, ncb.ctypes.data_as(C.POINTER(C.c_longlong)), len(blocks), m, int(ensize_K or 0),
             int(N_cluster or 0), int(minN_cluster or 0), int(maxN_cluster or 0),
             C.c_double(rN_seed), _ip(pred), C.byref(npred), C.byref(pu), _dp(viE))
    K = int(ensize_K or 5)
    out = _enresults(pred, None, None, n, m, pu.value, K, t0, {}, False, key="N.pred_clusters")
    if viewflag:                                                          # :215-232
        out["viE"] = viE
        out["x0"] = _one_hot(pred, npred.value)
    return out


def SHARP_unlimited2(scExp, ensize_K=None, reduced_ndim=None, partition_ncells=None, hmethod=None, N_cluster=None,
                     enpN_cluster=None, indN_cluster=None, minN_cluster=None, maxN_cluster=None, sil_thre=None,
                     height_Ntimes=None, logflag=None, n_cores=None, forview=True, rN_seed=None, testlog_cells=None):
    """R/SHARP_unlimited2.R:29-292 (with SHARP_fpart, :297-544): the list-of-blocks variant that takes log10, rounds the
    projections to one decimal before the base clustering and runs ONE sMetaC over the per-fold ensemble clusters of all
    blocks.  scExp: list of (genes, cells) blocks."""
    import time as _t

    t0 = _t.time()
    if scExp is None:
        raise SharpError("No expression data is provided!")
    if not isinstance(scExp, (list, tuple)):
        raise SharpError("The input should be a LIST of partitioned scRNA-seq expression matrices!")
    if rN_seed is not None:
        if not isinstance(rN_seed, (int, float, np.integer, np.floating)):
            raise SharpError("The rN.seed should be a numeric!")
        if rN_seed % 1 != 0:
            raise SharpError("The rN.seed should be an integer!")
    else:
        rN_seed = 0.5
    _lib.ensure_init()
    blocks = [np.asfortranarray(b, dtype=np.float64) for b in scExp]
    m = blocks[0].shape[0]
    ncb = np.array([b.shape[1] for b in blocks], np.int64)
    n = int(ncb.sum())
    p = int(reduced_ndim) if reduced_ndim else int(np.ceil(np.log2(n) / 0.04))          # :42-44
    if logflag is None:
        logflag = n < 1e4                                                               # :64-70
    flag = True
    if logflag:                                                                         # :71-82: testlog on the FIRST block
        nc1 = blocks[0].shape[1]
        flag = testlog(blocks[0], nc1, p, sncells=100, cells=testlog_cells)
    ptrs = (C.POINTER(C.c_double) * len(blocks))(*[_dp(b) for b in blocks])
    pred = np.zeros(n, np.int32)
    npred, pu = C.c_int(), C.c_int()
    viE = np.zeros((n, p)) if forview else None
    rc = check(lib().sharp_SHARP_unlimited2(ptrs, ncb.ctypes.data_as(C.POINTER(C.c_longlong)), len(blocks), m, int(ensize_K or 0),
                                            int(reduced_ndim or 0), int(partition_ncells or 0), _hmethod(hmethod),
                                            int(N_cluster or 0), int(enpN_cluster or 0), int(indN_cluster or 0),
                                            int(minN_cluster or 0), int(maxN_cluster or 0),
                                            C.c_double(-1.0 if sil_thre is None else sil_thre),
                                            C.c_double(0.0 if height_Ntimes is None else height_Ntimes), int(bool(flag)),
                                            C.c_double(rN_seed), _ip(pred), C.byref(npred), C.byref(pu), _dp(viE)), allow=48)
    K = int(ensize_K or 5)
    paras = {"ensize.K": K, "reduced.ndim": pu.value, "partition.ncells": int(partition_ncells or 2000), "logmark": bool(flag),
             "hmethod": hmethod or "ward.D", "N.cluster": N_cluster, "minN.cluster": int(minN_cluster or 2),
             "maxN.cluster": int(maxN_cluster or max(40, -(-n // 5000))), "sil.thre": 0.35 if sil_thre is None else sil_thre,
             "height.Ntimes": 2 if height_Ntimes is None else height_Ntimes, "n.cores": n_cores}
    out = _enresults(pred, None, None, n, m, pu.value, K, t0, paras, False, key="N.pred_clusters")
    out["reduced.ndim"] = pu.value                                                      # :226 (this variant names it so)
    if forview:
        out["viE"] = viE
        out["x0"] = _one_hot(pred, npred.value)
    out["warn"] = rc
    return out


def SHARP_unlimited3(ndinfo, viewflag=True, n_cores=None, ensize_K=None, rN_seed=None, N_cluster=None, minN_cluster=None,
                     maxN_cluster=None, logflag=False, testlog_cells=None, group=6):
    """R/SHARP_unlimited3.R:29-235: SHARP_unlimited over a DIRECTORY of partitions.

    ndinfo: dict(dir=..., ncells=..., ngenes=...) like the reference's list; the partitions are block files written by
    sharp_amd.blocks.write_block (the reference's .rds needs R to be read; dense float32 or the packed sparse format: 4 bytes per
    non-zero for counts), taken in the order of the first number in their path (:59-61) and streamed disk -> pinned memory -> HBM
    through a ring of buffers ahead of the clustering, which takes the blocks that have arrived together.
    logflag: False = log2 always on, as SHARP_unlimited passes it; True = leave it to testlog() per block, which is
    what unlimited3's SHARP() call does (:122; unseeded sample -> not reproducible unless testlog_cells is given).
    group (no reference counterpart): how many ARRIVED blocks the clustering takes together as one pipelined batch (1: block after block)."""
    import time as _t

    import torch

    from . import blocks as _blocks
    from . import device as _device

    t0 = _t.time()
    if ndinfo is None:
        raise SharpError("No expression data is provided!")
    try:
        files = _blocks.list_block_files(ndinfo["dir"])
    except FileNotFoundError as e:
        raise SharpError(str(e))
    ncells, ngenes = int(ndinfo["ncells"]), int(ndinfo["ngenes"])
    if rN_seed is not None:
        if not isinstance(rN_seed, (int, float, np.integer, np.floating)):
            raise SharpError("The rN.seed should be a numeric!")
        if rN_seed % 1 != 0:
            raise SharpError("The rN.seed should be an integer!")
    else:
        rN_seed = 0.5
    K = int(ensize_K or 5)
    p = int(np.ceil(np.log2(ncells) / 0.04))                                       # :66, from ndinfo$ncells
    _lib.ensure_init()
    proj = Projector(ngenes, p, [0.5 if rN_seed == 0.5 else 50 + rN_seed + k for k in range(1, K + 1)])   # :84-90
    preds, means, counts, views, nnc = [], [], [], [], []
    stream = None
    t_cluster = 0.0
    try:
        stream = _blocks.BlockStreamer(files)
        # The blocks that have arrived when the clustering asks for more go TOGETHER (up to `group`: one pipelined batch of base-clustering tasks,
        # sharp_unlimited_blocks_dev) -- unless their E1 rows are wanted or testlog has to look at each block, which the per-block entry serves.
        for grp in stream.groups(1 if (viewflag or logflag) else group):
            for i, hdr, dX in grp:
                if hdr["genes"] != ngenes:
                    raise SharpError("%s has %d genes, ndinfo$ngenes is %d" % (files[i], hdr["genes"], ngenes))
            t_c0 = _t.perf_counter()
            if len(grp) >= 2:
                for (i, hdr, dX), (pr, mn, cn) in zip(grp, _device.unlimited_blocks_dev([g[2] for g in grp], p, proj.handle, K, rN_seed)):
                    preds.append(pr); means.append(mn); counts.append(cn); views.append(None); nnc.append(hdr["cells"])
                t_cluster += _t.perf_counter() - t_c0
                continue
            i, hdr, dX = grp[0]
            nb = hdr["cells"]
            flag = True
            if logflag:                                                            # the block's SHARP() runs testlog (R/SHARP.R:205-222)
                cells = testlog_cells if testlog_cells is not None else np.random.default_rng().permutation(nb)[:100]
                cells = np.asarray(cells)[np.asarray(cells) < nb]
                sample = dX[torch.as_tensor(cells, device=dX.device)].cpu().numpy().T.astype(np.float64)
                flag = testlog(sample, sample.shape[1], p, cells=np.arange(sample.shape[1]))
            vi = np.zeros((nb, p)) if viewflag else None
            pr, mn, cn = _device.unlimited_block_dev(dX, p, proj.handle, K, rN_seed, flag=flag, viE=vi)
            preds.append(pr); means.append(mn); counts.append(cn); views.append(vi); nnc.append(nb)
            t_cluster += _t.perf_counter() - t_c0
    finally:
        proj.close()
        if stream is not None:
            stream.close()
    n = int(sum(nnc))
    first = np.concatenate([[0], np.cumsum([m_.shape[0] for m_ in means])])
    fid, nf = _device.unlimited_merge(np.concatenate(means, 0), np.concatenate(counts, 0), n, int(N_cluster or 0),
                                      int(minN_cluster or 0), int(maxN_cluster or 0))
    pred = np.concatenate([np.asarray(fid)[first[b] + preds[b] - 1] for b in range(len(preds))]).astype(np.int32)
    out = _enresults(pred, None, None, n, ngenes, p, K, t0, {}, False, key="N.pred_clusters")
    if viewflag:
        E1 = np.concatenate(views, 0)
        out["viE"] = _view_reduce(E1, rN_seed, K) if n > 1e5 else E1
        out["x0"] = _one_hot(pred, nf)
    out["bytes_streamed"] = stream.bytes_streamed
    out["read_seconds"] = stream.read_seconds              # the reader thread's file reads (wall), hidden under the clustering except for ...
    out["wait_seconds"] = stream.wait_seconds              # ... what the clustering waited for blocks that had not arrived
    out["expand_seconds"] = stream.expand_seconds          # packed blocks -> dense blocks on the device (before their clustering)
    out["cluster_seconds"] = t_cluster                     # the per-block / per-group library calls
    return out


def _view_reduce(E1, rN_seed, ensize_K, kdim=50):
    """R/SHARP_unlimited.R:217-225: above 1e5 cells viE = 1/sqrt(kdim) * E1 %*% ranM2(p, kdim, seed).

    The reference's seed expression `50 + rN.seed + k` reads a variable `k` that only exists inside the foreach at :96
    (an error in R whenever a seed is given); fixed here as the next seed of that sequence,
    k = ensize.K + 1."""
    p = E1.shape[1]
    seed = 0.5 if rN_seed == 0.5 else 50 + rN_seed + ensize_K + 1
    pr = Projector(p, kdim, [seed])
    try:
        return pr.project(np.ascontiguousarray(E1.T), logflag=False)     # (1/sqrt(kdim)) z0^T E1^T, cells x kdim
    finally:
        pr.close()


def _one_hot(pred, ncl):
    """x0 = sparseMatrix(i = 1:ncells, j = finalrowColor, x = 1) (R/SHARP_unlimited.R:230): CSR if scipy is there."""
    n = pred.size
    try:
        from scipy.sparse import csr_matrix

        # one entry per row: the CSR arrays written directly (the (row, col) form goes through a COO -> CSR conversion: 25 ms at 5e5 cells)
        return csr_matrix((np.ones(n), np.asarray(pred, np.int32) - 1, np.arange(n + 1, dtype=np.int32)), shape=(n, ncl))
    except Exception:  # pragma: no cover
        x0 = np.zeros((n, ncl))
        x0[np.arange(n), pred - 1] = 1.0
        return x0


def run_Mtimes_SHARP(scExp, Mtimes=10, Kset=15, **kwargs):
    """R/run_Mtimes_SHARP.R:20-60: SHARP(scExp, ensize.K = k, forview = FALSE, ...) Mtimes for every k of Kset.

    Returns {"enSize_<k>": {"Run_<j>": enresults}} like the reference's nested lists."""
    ks = [Kset] if np.isscalar(Kset) else list(Kset)
    allresults = {}
    for k in ks:
        info = {}
        for j in range(1, int(Mtimes) + 1):
            info["Run_%d" % j] = SHARP(scExp, ensize_K=int(k), forview=False, **kwargs)
        allresults["enSize_%d" % int(k)] = info
    return allresults


def get_marker_genes(scExp, y, theta=1e-4, auc=0.7, pvalue=0.01, FC=2, ng=1, n_cores=None, gene_names=None):
    """R/get_marker_genes.R:25-264.  scExp: (genes, cells); y: a SHARP() result (or anything with "pred_clusters").

    Returns dict(mginfo, gallinfo, mat, label, logmark): `gallinfo` has one row per gene that passed the sparsity / NaN
    filters (columns gene, auc, icluster, pvalue (Holm-adjusted), sparsity, FC); `mginfo` the selected marker genes ordered
    by (icluster, -FC, -auc, pvalue, -sparsity) like :177; `mat` their expression rows.  Tables are dicts of numpy columns."""
    _lib.ensure_init()
    X = np.asfortranarray(scExp, dtype=np.float64)
    m, n = X.shape
    names = np.arange(m) if gene_names is None else np.asarray(gene_names)
    keep = np.ones(m, bool)
    if gene_names is not None:                                            # :49-54 duplicated gene names are dropped
        _, first = np.unique(names, return_index=True)
        keep[:] = False
        keep[first] = True
        X, names = np.asfortranarray(X[keep]), names[keep]
        m = X.shape[0]
    pred = np.asarray(y["pred_clusters"] if isinstance(y, dict) else y)
    uy = np.unique(pred)                                                  # :95-99: use the index among the unique ids
    label = (np.searchsorted(uy, pred) + 1).astype(np.int32)
    G = int(uy.size)
    out = np.zeros((m, 5))
    check(lib().sharp_marker_genes(_dp(X), m, C.c_longlong(n), C.c_longlong(m), _ip(label), G, C.c_double(theta), int(ng), _dp(out)))
    cols = {"gene": names, "auc": out[:, 0], "icluster": out[:, 1].astype(np.int64), "pvalue": out[:, 2].copy(),
            "sparsity": out[:, 3], "FC": out[:, 4]}
    sel = (cols["sparsity"] > theta) & ~np.isnan(cols["pvalue"])          # :158-159
    g = {k: v[sel] for k, v in cols.items()}
    g["pvalue"] = _p_adjust_holm(g["pvalue"])                             # :160
    gall = {k: v.copy() for k, v in g.items()}
    if g["auc"].size:                                                     # :166-169: adauc = min(auc, min over clusters of max auc)
        maxauc = [g["auc"][g["icluster"] == c].max() for c in np.unique(g["icluster"])]
        adauc = min(auc, min(maxauc))
    else:
        adauc = auc
    pick = (g["pvalue"] < pvalue) & (g["auc"] > adauc) & (g["FC"] >= FC)  # :171-176
    s = {k: v[pick] for k, v in g.items()}
    order = np.lexsort((-s["sparsity"], s["pvalue"], -s["auc"], -s["FC"], s["icluster"]))   # :177
    s = {k: v[order] for k, v in s.items()}
    idx = {nm: i for i, nm in enumerate(names.tolist())}
    rows = np.array([idx[nm] for nm in s["gene"].tolist()], dtype=np.int64)
    return {"mginfo": s, "gallinfo": gall, "mat": X[rows] if rows.size else X[:0], "label": label,
            "logmark": (y.get("paras", {}) or {}).get("logmark") if isinstance(y, dict) else None}


def _marker_select(out, names, theta, auc, pvalue, ncols=4):
    """R/get_marker_genes_unlimited.R:131-146 (the same lines in _unlimited2 :193-214): sparsity / NaN filters, Holm adjustment,
    adauc = min(auc, min over clusters of the best auc), selection by adjusted p-value and auc."""
    cols = {"gene": names, "auc": out[:, 0], "icluster": out[:, 1].astype(np.int64), "pvalue": out[:, 2].copy(), "sparsity": out[:, 3]}
    sel = (cols["sparsity"] > theta) & ~np.isnan(cols["pvalue"])
    g = {k: v[sel] for k, v in cols.items()}
    g["pvalue"] = _p_adjust_holm(g["pvalue"])
    gall = {k: v.copy() for k, v in g.items()}
    if g["auc"].size:
        adauc = min(auc, min(g["auc"][g["icluster"] == c].max() for c in np.unique(g["icluster"])))
    else:
        adauc = auc
    pick = (g["pvalue"] < pvalue) & (g["auc"] > adauc)
    return {k: v[pick] for k, v in g.items()}, gall


def _marker_labels(y):
    pred = np.asarray(y["pred_clusters"] if isinstance(y, dict) else y)
    uy = np.unique(pred)                                                  # the index among the unique ids (R: match(y$pred_clusters, uy))
    return (np.searchsorted(uy, pred) + 1).astype(np.int32), int(uy.size)


def get_marker_genes_unlimited(scExp, y, theta=1e-5, auc=0.85, pvalue=0.01, n_cores=None, gene_names=None):
    """R/get_marker_genes_unlimited.R:25-187.  scExp: the LIST of (genes, cells) blocks SHARP_unlimited clustered (dense arrays or
    scipy.sparse matrices standing in for the reference's dgCMatrix blocks); y: its result.  Genes that are zero in every block are
    dropped (:44-57); per remaining gene the values across all blocks are ranked over ALL cells, the cluster of highest mean rank
    gets the Wilcoxon p-value and the AUROC (:95-118; sharp_marker_genes_blocks_*: one pass per block into per-gene lists of non-zero
    cells, one segmented sort, one statistics pass); Holm adjustment and selection :131-146.
    Returns dict(mginfo, mat, label): mat = the selected genes' rows across all cells (:151-155)."""
    _lib.ensure_init()
    if not isinstance(scExp, (list, tuple)) or len(scExp) < 1:
        raise SharpError("The input should be a LIST of partitioned scRNA-seq expression matrices!")
    label, G = _marker_labels(y)
    sparse = all(_is_sparse(b) for b in scExp)
    m = scExp[0].shape[0]
    ncb = np.array([b.shape[1] for b in scExp], np.int64)
    if int(ncb.sum()) != label.size:
        raise SharpError("get_marker_genes_unlimited: the labels do not match the blocks' cells")
    names = np.arange(m) if gene_names is None else np.asarray(gene_names)
    out = np.zeros((m, 5))
    B = len(scExp)
    if sparse:
        blocks = [b.tocsc() for b in scExp]
        for b in blocks:
            b.sum_duplicates()
        cps, ris = _csc_int_slots(blocks)
        vxs = [np.ascontiguousarray(b.data, np.float64) for b in blocks]
        cpp = (C.POINTER(C.c_int) * B)(*[_ip(v) for v in cps])
        rip = (C.POINTER(C.c_int) * B)(*[_ip(v) if v.size else C.cast(None, C.POINTER(C.c_int)) for v in ris])
        vxp = (C.POINTER(C.c_double) * B)(*[_dp(v) if v.size else C.cast(None, C.POINTER(C.c_double)) for v in vxs])
        check(lib().sharp_marker_genes_blocks_csc(cpp, rip, vxp, ncb.ctypes.data_as(C.POINTER(C.c_longlong)), B, m, _ip(label), G,
                                                  C.c_double(theta), 1, _dp(out)))
    else:
        import torch

        dbl = [torch.from_numpy(np.ascontiguousarray(np.asarray(b.toarray() if _is_sparse(b) else b).T.astype(np.float32))).cuda() for b in scExp]
        torch.cuda.synchronize()
        ptrs = (C.c_void_p * B)(*[t.data_ptr() for t in dbl])
        ldb = np.array([t.stride(0) for t in dbl], np.int64)
        check(lib().sharp_marker_genes_blocks_dev(ptrs, ncb.ctypes.data_as(C.POINTER(C.c_longlong)), ldb.ctypes.data_as(C.POINTER(C.c_longlong)),
                                                  B, m, _ip(label), G, C.c_double(theta), 1, _dp(out)))
    nonzero = out[:, 3] > 0                                               # :44-57: a gene that is zero in every block is not a row of g5 at all
    s, _gall = _marker_select(out[nonzero], names[nonzero], theta, auc, pvalue)
    rows = np.array([int(np.nonzero(names == nm)[0][0]) for nm in s["gene"].tolist()], dtype=np.int64)
    if rows.size:
        mat = np.concatenate([np.asarray(b[rows].toarray() if _is_sparse(b) else np.asarray(b)[rows], dtype=np.float64) for b in scExp], axis=1)
    else:
        mat = np.zeros((0, label.size))
    return {"mginfo": s, "mat": mat, "label": label}


def get_marker_genes_unlimited2(gdinfo, y, theta=1e-5, auc=0.85, pvalue=0.05, n_cores=None):
    """R/get_marker_genes_unlimited2.R:25-411.  gdinfo: a directory of GENE-wise partitions -- file i holds some genes' values over ALL
    cells (the reference: .rds matrices, genes x cells; here the block files of sharp_amd.blocks.write_block, a (genes_i, cells) array
    each), taken in the order of the first number in their path (:139-142).  Per gene: rank over all cells, the min(10, N.cluster)
    clusters of highest mean rank tried, the best AUROC kept with its Wilcoxon p-value (:152-190: sharp_marker_genes per file, which is
    the same per-gene pass with ng = min(10, N.cluster)); Holm adjustment and selection :193-214.
    Returns dict(mginfo, label, gallinfo)."""
    from . import blocks as _blocks

    _lib.ensure_init()
    label, G = _marker_labels(y)
    n = label.size
    files = _blocks.list_block_files(gdinfo)
    rr = min(10, G)
    outs, names = [], []
    for f in files:
        X = _blocks.read_block(f)                                         # (genes_i, cells)
        if X.shape[1] != n:
            raise SharpError("get_marker_genes_unlimited2: %s holds %d cells, the labels %d" % (f, X.shape[1], n))
        Xf = np.asfortranarray(X, dtype=np.float64)
        mi = Xf.shape[0]
        o = np.zeros((mi, 5))
        check(lib().sharp_marker_genes(_dp(Xf), mi, C.c_longlong(n), C.c_longlong(mi), _ip(label), G, C.c_double(theta), rr, _dp(o)))
        outs.append(o)
        names.extend("%s:%d" % (os.path.basename(f), j) for j in range(mi))
    out = np.concatenate(outs) if outs else np.zeros((0, 5))
    s, gall = _marker_select(out, np.asarray(names), theta, auc, pvalue)
    return {"mginfo": s, "label": label, "gallinfo": gall}


def _p_adjust_holm(p):
    """stats::p.adjust(p, "holm"): pmin(1, cummax((n - i + 1) * p[o]))[ro]."""
    p = np.asarray(p, np.float64)
    n = p.size
    if n == 0:
        return p
    o = np.argsort(p, kind="stable")
    adj = np.minimum(1.0, np.maximum.accumulate((n - np.arange(n)) * p[o]))
    out = np.empty(n)
    out[o] = adj
    return out


DECISION_COLS = 14
DECISION_FIELDS = ("level", "block", "k", "fold", "n", "branch", "chosen_k", "ties", "best", "runner_up", "sil_minus_thre", "height_ratio",
                   "smetac_override_k", "levels")


def decision_log(enable=True):
    """sharp_decision_log: while on, every get_opt_hclust call of the process -- base clustering, wMetaC, sMetaC,
    the cross-block merge -- leaves one row saying which rule of R/get_opt_hclust.R:162-229 chose the number of clusters and by what
    margin.  Switching it (on or off) clears the log.  SHARP_DECISION_LOG=1 in the environment: on from the start."""
    _lib.ensure_init()
    check(lib().sharp_decision_log(int(bool(enable))))


def last_decisions():
    """sharp_last_decisions: rows x DECISION_COLS (the columns of DECISION_FIELDS; include/sharp_hip.h), sorted by (level, block, k, fold)."""
    n = C.c_int()
    check(lib().sharp_last_decisions(None, 0, C.byref(n)))
    rows = np.zeros((max(n.value, 1), DECISION_COLS))
    check(lib().sharp_last_decisions(_dp(rows), rows.shape[0], C.byref(n)))
    return rows[: n.value]


def decision_margins(rows):
    """Per level: how close the decisions of a run came to going the other way.  For a decision by the median silhouette with one exact
    maximum the margin is best - runner-up; with exact ties (R picks the middle one by `==` on doubles, R/get_opt_hclust.R:162-168) the
    margin is 0 by construction and the ties are counted instead; a decision by CH: the relative margin (best - runner-up) / |best|;
    every decision also has its distance to the silhouette / CH switch, |max(msil) - sil.thre|.  -> {level: {...}}"""
    out = {}
    names = {0: "base", 1: "wMetaC", 2: "sMetaC", 3: "merge", -1: "direct"}
    for lev in sorted(set(rows[:, 0].astype(int))) if len(rows) else []:
        r = rows[rows[:, 0] == lev]
        sil = r[r[:, 5] == 0]
        ch = r[(r[:, 5] == 1) | (r[:, 5] == 2)]
        single = sil[sil[:, 7] == 1]
        d = {"decisions": int(len(r)), "by_silhouette": int(len(sil)), "by_CH": int((r[:, 5] == 1).sum()), "by_height": int((r[:, 5] == 2).sum()),
             "N_cluster_given": int((r[:, 5] == 3).sum()),
             "silhouette_decisions_with_exact_ties": int((sil[:, 7] > 1).sum()),
             "min_silhouette_margin": float(np.nanmin(single[:, 8] - single[:, 9])) if len(single) and np.isfinite(single[:, 9]).any() else None,
             "min_CH_relative_margin": float(np.nanmin((ch[:, 8] - ch[:, 9]) / np.abs(ch[:, 8]))) if len(ch) and np.isfinite(ch[:, 9]).any() else None,
             "min_distance_to_sil_thre": float(np.nanmin(np.abs(r[:, 10]))) if np.isfinite(r[:, 10]).any() else None,
             "smetac_overrides": int((r[:, 12] > 0).sum())}
        out[names.get(lev, str(lev))] = d
    return out


def ARI(label, res):
    """R/ARI.R:20-42: clues::adjustedRand(label, res$pred_clusters) -> Rand, HA, MA, FM, Jaccard."""
    a = _labels_to_int(np.asarray(label))
    b = _labels_to_int(np.asarray(res["pred_clusters"] if isinstance(res, dict) else res))
    n = a.size
    tab = np.zeros((a.max() + 1, b.max() + 1))
    np.add.at(tab, (a, b), 1)
    ra, rb = tab.sum(1), tab.sum(0)
    sij = (tab * (tab - 1) / 2).sum()
    si = (ra * (ra - 1) / 2).sum()
    sj = (rb * (rb - 1) / 2).sum()
    tot = n * (n - 1) / 2
    A, B, Cc = sij, si - sij, sj - sij
    D = tot - A - B - Cc
    rand = (A + D) / tot
    e = si * sj / tot
    ha = (sij - e) / (0.5 * (si + sj) - e)
    erand = (tot + (ra ** 2).sum() * (rb ** 2).sum() / n ** 2 - 0.5 * ((ra ** 2).sum() + (rb ** 2).sum())) / tot
    return {"Rand": rand, "HA": ha, "MA": (rand - erand) / (1 - erand), "FM": A / np.sqrt((A + B) * (A + Cc)),
            "Jaccard": A / (A + B + Cc)}
