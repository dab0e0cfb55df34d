"""Host-side mirror of the reference's R interface for the hot path.

Function names, argument names/meaning and error behaviour follow the R package
(dots in R argument names become underscores).  Everything is computed by
libsharp_hip.so through ctypes; arrays are numpy on the host boundary."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import SharpError, check, lib

__all__ = ["ranM", "ranM2", "RPmat", "Projector", "SharpError", "get_opt_hclust", "getrowColor", "colorL", "HMETHODS"]


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
    """R/ranM2.R:44-68 -- as ranM but takes the number of features."""
    _seed_check(seedn)
    return Projector(m, p, [seedn])


def RPmat(scdata, p, seedn):
    """R/RPmat.R:82-115 -- list(R = projector, projmat = 1/sqrt(p) * t(R) %*% scdata) (p x n)."""
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


def _hmethod(h):
    if h is None:
        return 1
    if h not in HMETHODS:
        raise SharpError(f"invalid clustering method '{h}'")
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
    rc = check(lib().sharp_get_opt_hclust(_dp(a), n, p, _hmethod(hmethod), Ncl, minN, maxN, C.c_double(sil),
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
    check(lib().sharp_getrowColor(_dp(a), n, p, _hmethod(hmethod), int(indN_cluster or 0), int(minN_cluster),
                                  int(maxN_cluster), C.c_double(sil_thre),
                                  C.c_double(1.0 if height_Ntimes is None else height_Ntimes), _ip(rc_), C.byref(maxsil)),
          allow=16)
    return {"rowColor": [colorL[j - 1] for j in rc_], "rowColor_id": rc_, "maxsil": maxsil.value, "mat": Emat}
