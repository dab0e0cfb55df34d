"""Host-side mirror of the reference's R interface for the hot path.

Function names, argument names/meaning and error behaviour follow the R package
(dots in R argument names become underscores).  Everything is computed by
libsharp_hip.so through ctypes; arrays are numpy on the host boundary."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import SharpError, check, lib

__all__ = ["ranM", "ranM2", "RPmat", "Projector", "SharpError"]


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
