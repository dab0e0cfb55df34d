"""Builds r/sharp_glue.c (this repository's .Call shim) against tests/rmock (a stand-in for the few R C-API functions it uses: the image
has no R) and wraps it for ctypes.  Test infrastructure."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tests", "rmock", "libsharp_glue_mock.so")


def build():
    src = [os.path.join(ROOT, "r", "sharp_glue.c"), os.path.join(ROOT, "tests", "rmock", "rmock.c")]
    deps = src + [os.path.join(ROOT, "tests", "rmock", "Rinternals.h"), os.path.join(ROOT, "include", "sharp_hip.h")]
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        lib_dir = os.path.join(ROOT, "sharp_amd")
        subprocess.check_call(["gcc", "-O1", "-Wall", "-Werror", "-Wno-cast-function-type", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "tests", "rmock"),
                               "-I" + os.path.join(ROOT, "include")] + src + ["-o", SO, "-L" + lib_dir, "-lsharp_hip",
                                                                                "-Wl,-rpath," + lib_dir, "-lm"])
    return SO


class Glue:
    def __init__(self):
        import sharp_amd

        sharp_amd.lib()                                 # libsharp_hip.so (and torch's HIP runtime) first: one runtime per process
        self.L = L = C.CDLL(build())
        S = C.c_void_p
        for name, res, args in [("rmock_real_matrix", S, [S, C.c_int, C.c_int]), ("rmock_real_vector", S, [S, C.c_ssize_t]),
                                ("rmock_int_vector", S, [S, C.c_ssize_t]), ("rmock_logical", S, [C.c_int]), ("rmock_list", S, [C.c_int]),
                                ("rmock_list_set", None, [S, C.c_int, S, C.c_char_p]), ("rmock_list_get", S, [S, C.c_char_p]),
                                ("rmock_call", S, [S, C.c_int, C.POINTER(S)]), ("rmock_last_error", C.c_char_p, []),
                                ("rmock_warnings", C.c_char_p, []), ("rmock_protect_depth", C.c_int, []), ("rmock_registered", C.c_int, [C.c_char_p]),
                                ("rmock_reset", None, []), ("rmock_type", C.c_int, [S]), ("rmock_data", S, [S]), ("XLENGTH", C.c_ssize_t, [S]),
                                ("Rf_nrows", C.c_int, [S]), ("Rf_ncols", C.c_int, [S]), ("R_init_sharp_glue", None, [S])]:
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        L.R_init_sharp_glue(None)

    # ---- R values
    def matrix(self, a):                                # a: (nrow, ncol) array -> R numeric matrix (column-major)
        a = np.asfortranarray(a, np.float64)
        return self.L.rmock_real_matrix(a.ctypes.data, a.shape[0], a.shape[1])

    def real(self, *v):
        a = np.ascontiguousarray(np.array(v, np.float64).ravel())
        return self.L.rmock_real_vector(a.ctypes.data, a.size)

    def int(self, *v):
        a = np.ascontiguousarray(np.array(v, np.int32).ravel())
        return self.L.rmock_int_vector(a.ctypes.data, a.size)

    def lgl(self, v):
        return self.L.rmock_logical(int(bool(v)))

    def list(self, items, names=None):
        lst = self.L.rmock_list(len(items))
        for i, it in enumerate(items):
            self.L.rmock_list_set(lst, i, it, names[i].encode() if names else None)
        return lst

    def csc_block(self, sp):
        """what r/sharp_hip.R::.sharp_block makes of a Matrix::dgCMatrix: list(p = @p, i = @i, x = @x, dim = @Dim)"""
        sp = sp.tocsc()
        return self.list([self.int(sp.indptr), self.int(sp.indices), self.real(sp.data), self.int(*sp.shape)], ["p", "i", "x", "dim"])

    # ---- .Call
    def call(self, name, *args):
        """.Call(name, ...): the result SEXP, or raises RuntimeError with the message R's error() would carry"""
        assert self.L.rmock_registered(name.encode()) == len(args), "argument count differs from the registration table"
        arr = (C.c_void_p * max(1, len(args)))(*args)
        r = self.L.rmock_call(C.cast(getattr(self.L, name), C.c_void_p), len(args), arr)
        assert self.L.rmock_protect_depth() == 0, "PROTECT / UNPROTECT imbalance"
        if not r:
            raise RuntimeError(self.L.rmock_last_error().decode())
        return r

    def get(self, lst, name):
        return self.value(self.L.rmock_list_get(lst, name.encode()))

    def value(self, s):
        t, n = self.L.rmock_type(s), self.L.XLENGTH(s)
        if t == 0:
            return None
        ct = C.c_double if t == 14 else C.c_int
        a = np.ctypeslib.as_array(C.cast(self.L.rmock_data(s), C.POINTER(ct)), (max(n, 1),))[:n].copy()
        nr, nc = self.L.Rf_nrows(s), self.L.Rf_ncols(s)
        if nr * nc == n and nc > 1:
            a = a.reshape(nc, nr).T                     # column-major R matrix -> (nrow, ncol)
        return a

    def warnings(self):
        return self.L.rmock_warnings().decode()

    def reset(self):
        self.L.rmock_reset()
