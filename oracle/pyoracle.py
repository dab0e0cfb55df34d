"""ctypes binding of the CPU oracle (oracle/sharp_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under sharp_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    src = os.path.join(_HERE, "sharp_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.oracle_median_silhouette.restype = C.c_double
        _lib.oracle_get_CH_1corr.restype = C.c_double
        _lib.oracle_synth_value.restype = C.c_float
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _dp(a):
    return None if a is None else _p(a, C.c_double)


def _ip(a):
    return None if a is None else _p(a, C.c_int)


HMETHODS = {"ward.D": 1, "single": 2, "complete": 3, "average": 4, "mcquitty": 5,
            "median": 6, "centroid": 7, "ward.D2": 8}


def runif(seed, n):
    out = np.empty(n, np.float64)
    lib().oracle_runif(C.c_uint32(seed), n, _dp(out))
    return out


def sample_perm(seed, n):
    out = np.empty(n, np.int32)
    lib().oracle_sample_perm(C.c_uint32(seed), n, _ip(out))
    return out


def ranM(m, p, seedn):
    """ternary projector, (m, p) int8 in {+1,0,-1}; magnitude is sqrt(sqrt(m))."""
    t = np.empty((m, p), np.int8)
    lib().oracle_ranM(m, p, C.c_double(seedn), _p(t, C.c_int8))
    return t


def project(X, tern, logflag=True):
    """X: (m, n) genes x cells (any layout; copied to column-major). Returns E (n, p)."""
    m, n = X.shape
    Xf = np.asfortranarray(X, dtype=np.float64)
    p = tern.shape[1]
    tern = np.ascontiguousarray(tern, dtype=np.int8)
    E = np.empty((n, p), np.float64)
    lib().oracle_project(_dp(Xf), m, n, _p(tern, C.c_int8), p, int(bool(logflag)), _dp(E))
    return E


def hclust(dcond, n, method="ward.D"):
    d = np.array(dcond, np.float64, copy=True)
    ia = np.zeros(n, np.int32)
    ib = np.zeros(n, np.int32)
    crit = np.zeros(n, np.float64)
    lib().oracle_hclust(n, HMETHODS[method], _dp(d), _ip(ia), _ip(ib), _dp(crit))
    return ia[: n - 1], ib[: n - 1], crit[: n - 1]


def scale_rows(mat):
    a = np.array(mat, np.float64, order="C", copy=True)
    lib().oracle_scale_rows(_dp(a), a.shape[0], a.shape[1])
    return a


def cor_dist(mat):
    a = np.ascontiguousarray(mat, np.float64)
    n, p = a.shape
    d = np.empty(n * (n - 1) // 2, np.float64)
    lib().oracle_cor_dist(_dp(a), n, p, _dp(d))
    return d


def silhouette_widths(cl, dcond):
    cl = np.ascontiguousarray(cl, np.int32)
    n = cl.size
    si = np.empty(n, np.float64)
    d = np.ascontiguousarray(dcond, np.float64)
    lib().oracle_silhouette_widths(n, int(cl.max()), _ip(cl), _dp(d), _dp(si))
    return si


def get_CH_1corr(y, cl):
    y = np.ascontiguousarray(y, np.float64)
    cl = np.ascontiguousarray(cl, np.int32)
    return lib().oracle_get_CH_1corr(_dp(y), y.shape[0], y.shape[1], _ip(cl), int(cl.max()))


def get_opt_hclust(mat, hmethod="ward.D", N_cluster=0, minN=2, maxN=40, sil_thre=0.35,
                   height_Ntimes=2.0):
    a = np.ascontiguousarray(mat, np.float64)
    n, p = a.shape
    nk = max(1, min(maxN, n - 1) - minN + 1)
    f = np.zeros(n, np.int32)
    v = np.zeros(n * nk, np.int32)
    msil = np.zeros(nk)
    ch = np.zeros(nk)
    height = np.zeros(n)
    maxsil = C.c_double()
    optN = C.c_int()
    nko = C.c_int()
    br = C.c_int()
    rc = lib().oracle_get_opt_hclust(_dp(a), n, p, HMETHODS[hmethod], int(N_cluster or 0), minN, maxN,
                                     C.c_double(sil_thre), C.c_double(height_Ntimes), _ip(f), _ip(v),
                                     _dp(msil), _dp(ch), C.byref(maxsil), _dp(height), C.byref(optN),
                                     C.byref(nko), C.byref(br))
    k = nko.value
    return dict(rc=rc, f=f, v=v[: n * k].reshape(k, n).T.copy(), msil=msil[:k], CHind=ch[:k],
                maxsil=maxsil.value, height=height[: n - 1], optN=optN.value, branch=br.value)


def getrowColor(E, hmethod="ward.D", indN=0, minN=2, maxN=40, sil_thre=0.35, height_Ntimes=1.0):
    a = np.ascontiguousarray(E, np.float64)
    n, p = a.shape
    rc_ = np.zeros(n, np.int32)
    maxsil = C.c_double()
    rc = lib().oracle_getrowColor(_dp(a), n, p, HMETHODS[hmethod], int(indN or 0), minN, maxN,
                                  C.c_double(sil_thre), C.c_double(height_Ntimes), _ip(rc_), C.byref(maxsil))
    return dict(rc=rc, rowColor=rc_, maxsil=maxsil.value)


def wMetaC(nC, hmethod="ward.D", enN=0, minN=2, maxN=40, sil_thre=0.35, height_Ntimes=2.0):
    """nC: (N, C) int labels. Returns finalC, x0, and debug intermediates."""
    a = np.asfortranarray(nC, dtype=np.int32)
    N, Cc = a.shape
    finalC = np.zeros(N, np.int32)
    cap = max(maxN, 2) + 2
    x0 = np.zeros(N * cap)
    w1 = np.zeros(N)
    allc_cap = N * Cc
    S = np.zeros(min(allc_cap, 4096) ** 2)
    tf = np.zeros(min(allc_cap, 4096), np.int32)
    ncl = C.c_int()
    allC = C.c_int()
    rc = lib().oracle_wMetaC(_ip(a), N, Cc, HMETHODS[hmethod], int(enN or 0), minN, maxN,
                             C.c_double(sil_thre), C.c_double(height_Ntimes), _ip(finalC), _dp(x0),
                             C.byref(ncl), _dp(w1), _dp(S), C.byref(allC), _ip(tf))
    A = allC.value
    return dict(rc=rc, finalC=finalC, x0=x0[: N * ncl.value].reshape(ncl.value, N).T.copy(), w1=w1,
                S=S[: A * A].reshape(A, A).copy(), tf=tf[:A].copy(), allC=A)


def sMetaC(labels, sE1, hmethod="ward.D", finalN=0, minN=2, maxN=40, sil_thre=0.35, height_Ntimes=2.0):
    lab = np.ascontiguousarray(labels, np.int32)
    E = np.ascontiguousarray(sE1, np.float64)
    n, p = E.shape
    fin = np.zeros(n, np.int32)
    tf = np.zeros(n, np.int32)
    nC = C.c_int()
    rc = lib().oracle_sMetaC(_ip(lab), _dp(E), n, p, HMETHODS[hmethod], int(finalN or 0), minN, maxN,
                             C.c_double(sil_thre), C.c_double(height_Ntimes), _ip(fin), _ip(tf), C.byref(nC))
    return dict(rc=rc, finalColor=fin, tf=tf[: nC.value].copy(), nC=nC.value)


def make_folds(ncells, ng=2000):
    f = np.zeros(ncells, np.int32)
    T = lib().oracle_make_folds(ncells, ng, _ip(f))
    return f, T


def SHARP_small(X, K=15, p=None, hmethod="ward.D", N_cluster=0, indN=0, minN=2, maxN=40, sil_thre=0.35,
                height_Ntimes=2.0, flag=True, rN_seed=2103, want_view=True):
    m, n = X.shape
    Xf = np.asfortranarray(X, dtype=np.float64)
    if p is None:
        p = int(np.ceil(np.log2(n) / 0.04))
    pred = np.zeros(n, np.int32)
    viE = np.zeros((n, p)) if want_view else None
    enrp = np.zeros((K, n), np.int32)
    x0 = np.zeros(n * (maxN + 2))
    ncl = C.c_int()
    rc = lib().oracle_SHARP_small(_dp(Xf), m, n, K, p, HMETHODS[hmethod], int(N_cluster or 0), int(indN or 0),
                                  minN, maxN, C.c_double(sil_thre), C.c_double(height_Ntimes), int(bool(flag)),
                                  C.c_double(rN_seed), _ip(pred), _dp(viE), _ip(enrp), _dp(x0), C.byref(ncl))
    return dict(rc=rc, pred_clusters=pred, viE=viE, enrp=enrp.T.copy(),
                x0=x0[: n * ncl.value].reshape(ncl.value, n).T.copy())


def SHARP(X, K=0, reduced_ndim=0, base_ncells=0, partition_ncells=0, hmethod="ward.D", N_cluster=0, enpN=0,
          indN=0, minN=0, maxN=0, sil_thre=-1.0, height_Ntimes=0.0, flag=True, tern=None, rN_seed=2103,
          nthreads=1, want_view=True):
    m, n = X.shape
    Xf = np.asfortranarray(X, dtype=np.float64)
    p = reduced_ndim if reduced_ndim > 0 else int(np.ceil(np.log2(n) / 0.04))
    pred = np.zeros(n, np.int32)
    viE = np.zeros((n, p)) if want_view else None
    pt = None if tern is None else np.ascontiguousarray(tern, np.int8)
    po = C.c_int()
    ko = C.c_int()
    rc = lib().oracle_SHARP(_dp(Xf), m, n, K, reduced_ndim, base_ncells, partition_ncells, HMETHODS[hmethod],
                            int(N_cluster or 0), int(enpN or 0), int(indN or 0), minN, maxN, C.c_double(sil_thre),
                            C.c_double(height_Ntimes), int(bool(flag)),
                            None if pt is None else _p(pt, C.c_int8), C.c_double(rN_seed), nthreads,
                            _ip(pred), _dp(viE), C.byref(po), C.byref(ko))
    return dict(rc=rc, pred_clusters=pred, viE=viE, p=po.value, K=ko.value)


def SHARP_large(X, K=5, p=None, ng=2000, hmethod="ward.D", N_cluster=0, enpN=0, indN=0, minN=2, maxN=0, sil_thre=0.35,
                height_Ntimes=2.0, flag=True, rN_seed=2103, nthreads=1):
    """R/SHARP.R:478-851 with its whole return contract: pred_clusters, viE and the soft cluster matrix x0 (:717-731,761-779)."""
    m, n = X.shape
    Xf = np.asfortranarray(X, dtype=np.float64)
    if p is None:
        p = int(np.ceil(np.log2(n) / 0.04))
    if maxN <= 0:
        maxN = max(40, -(-n // 5000))
    pred = np.zeros(n, np.int32)
    viE = np.zeros((n, p))
    cap = max(maxN, 40) + 2
    T = -(-n // ng)
    cap = max(cap, T * K * max(maxN, 40) + 2) if T == 1 else cap + T * 2
    x0 = np.zeros(n * cap)
    ncol = C.c_int()
    rc = lib().oracle_SHARP_large_x0(_dp(Xf), m, n, K, p, ng, HMETHODS[hmethod], int(N_cluster or 0), int(enpN or 0), int(indN or 0),
                                     minN, maxN, C.c_double(sil_thre), C.c_double(height_Ntimes), int(bool(flag)), None,
                                     C.c_double(rN_seed), nthreads, _ip(pred), _dp(viE), _dp(x0), cap, C.byref(ncol))
    return dict(rc=rc, pred_clusters=pred, viE=viE, x0=x0[: n * ncol.value].reshape(ncol.value, n).T.copy())


def SHARP_unlimited(blocks, K=0, N_cluster=0, minN=0, maxN=0, rN_seed=2103, nthreads=1, want_view=False):
    m = blocks[0].shape[0]
    ncb = np.array([b.shape[1] for b in blocks], np.int32)
    Xcat = np.concatenate([np.asfortranarray(b, dtype=np.float64).ravel(order="F") for b in blocks])
    n = int(ncb.sum())
    p = int(np.ceil(np.log2(n) / 0.04))
    pred = np.zeros(n, np.int32)
    viE = np.zeros((n, p)) if want_view else None
    po = C.c_int()
    rc = lib().oracle_SHARP_unlimited(_dp(Xcat), m, len(blocks), _ip(ncb), K, int(N_cluster or 0), minN, maxN,
                                      C.c_double(rN_seed), nthreads, _ip(pred), _dp(viE), C.byref(po))
    return dict(rc=rc, pred_clusters=pred, viE=viE, p=po.value)


def unlimited_merge(means, counts, ncells, N_cluster=0, minN=0, maxN=0):
    """The tail of SHARP_unlimited from the per-(block, cluster) centroids on (R/SHARP_unlimited.R:163-183)."""
    means = np.ascontiguousarray(means, np.float64)
    counts = np.ascontiguousarray(counts, np.int64)
    nC, p = means.shape
    fid = np.zeros(nC, np.int32)
    nf = C.c_int()
    rc = lib().oracle_unlimited_merge(_dp(means), _p(counts, C.c_longlong), nC, p, C.c_longlong(int(ncells)), int(N_cluster or 0),
                                      int(minN), int(maxN), _ip(fid), C.byref(nf))
    return dict(rc=rc, final_id=fid, n_final=nf.value)


def SHARP_unlimited2(blocks, K=0, reduced_ndim=0, partition_ncells=0, hmethod="ward.D", N_cluster=0, enpN=0, indN=0, minN=0,
                     maxN=0, sil_thre=-1.0, height_Ntimes=0.0, flag=True, rN_seed=2103, nthreads=1, want_view=False):
    m = blocks[0].shape[0]
    ncb = np.array([b.shape[1] for b in blocks], np.int32)
    Xcat = np.concatenate([np.asfortranarray(b, dtype=np.float64).ravel(order="F") for b in blocks])
    n = int(ncb.sum())
    p = int(reduced_ndim) if reduced_ndim else int(np.ceil(np.log2(n) / 0.04))
    pred = np.zeros(n, np.int32)
    viE = np.zeros((n, p)) if want_view else None
    po = C.c_int()
    rc = lib().oracle_SHARP_unlimited2(_dp(Xcat), m, len(blocks), _ip(ncb), K, int(reduced_ndim), int(partition_ncells),
                                       HMETHODS[hmethod], int(N_cluster or 0), int(enpN or 0), int(indN or 0), minN, maxN,
                                       C.c_double(sil_thre), C.c_double(height_Ntimes), int(bool(flag)), C.c_double(rN_seed),
                                       nthreads, _ip(pred), _dp(viE), C.byref(po))
    return dict(rc=rc, pred_clusters=pred, viE=viE, p=po.value)


STAGES = ("projector_build", "rp_matmul_thread_s", "base_clustering_thread_s", "wMetaC", "sMetaC_in_block", "cross_block_merge", "task_loop_wall")


DECISION_COLS = 14
DECISION_FIELDS = ("level", "block", "k", "fold", "n", "branch", "chosen_k", "ties", "best", "runner_up", "sil_minus_thre", "height_ratio",
                   "smetac_override_k", "levels")


def decision_log(enable=True):
    """oracle_decision_log: every get_opt_hclust call leaves a row until switched off; switching clears the log."""
    lib().oracle_decision_log(int(bool(enable)))


def last_decisions():
    """rows x DECISION_COLS, sorted by (level, block, k, fold): the columns of DECISION_FIELDS (include/sharp_hip.h, sharp_last_decisions)"""
    n = lib().oracle_last_decisions(None, 0)
    rows = np.zeros((max(n, 1), DECISION_COLS))
    n = lib().oracle_last_decisions(_dp(rows), rows.shape[0])
    return rows[:n]


def stage_seconds(reset=True):
    """Seconds per stage of the SHARP_large / SHARP_unlimited calls since the last reset (oracle_stage_seconds): bench.py's cpu_baseline."""
    out = np.zeros(7)
    lib().oracle_stage_seconds(_dp(out), int(bool(reset)))
    return dict(zip(STAGES, [round(float(v), 3) for v in out]))


def marker_genes(X, label, G, theta=1e-4, ng=1):
    """Per-gene (auc, icluster, pvalue, sparsity, FC) of R/get_marker_genes.R:120-152; X genes x cells."""
    X = np.asfortranarray(X, dtype=np.float64)
    m, n = X.shape
    lab = np.ascontiguousarray(label, np.int32)
    out = np.zeros((m, 5))
    lib().oracle_marker_genes(_dp(X), m, n, _ip(lab), int(G), C.c_double(theta), int(ng), _dp(out))
    return out


def round1(x):
    lib().oracle_round1.restype = C.c_double
    return np.array([lib().oracle_round1(C.c_double(float(v))) for v in np.ravel(x)]).reshape(np.shape(x))


def testlog(X, p, cells):
    m, n = X.shape
    Xf = np.asfortranarray(X, dtype=np.float64)
    cells = np.ascontiguousarray(cells, np.int32)
    ms = np.zeros(2)
    flag = lib().oracle_testlog(_dp(Xf), m, n, p, _ip(cells), cells.size, _dp(ms))
    return bool(flag), ms


def adjusted_rand(a, b):
    a = np.ascontiguousarray(a, np.int32)
    b = np.ascontiguousarray(b, np.int32)
    out = np.zeros(5)
    lib().oracle_adjusted_rand(_ip(a), _ip(b), a.size, _dp(out))
    return dict(Rand=out[0], HA=out[1], MA=out[2], FM=out[3], Jaccard=out[4])


def synth_fill(seed, m, cell0, ncell, G=12, nmark=1000):
    X = np.empty((m, ncell), np.float64, order="F")
    lib().oracle_synth_fill(C.c_uint32(seed), m, cell0, ncell, G, nmark, _dp(X))
    return X


def synth_cluster(seed, cells, G=12):
    return np.array([lib().oracle_synth_cluster(C.c_uint32(seed), C.c_uint32(int(c)), G) for c in cells], np.int32)
