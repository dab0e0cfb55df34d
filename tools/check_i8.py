"""The sliced-integer distance GEMM against an extended-precision numpy product and against the fp64 MFMA kernel."""
import sys, ctypes as C
import numpy as np
sys.path.insert(0, "/root/repo")
import sharp_amd
sharp_amd.init(0); lib = sharp_amd.lib()
dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
rng = np.random.default_rng(7)
for n, p in ((300, 50), (2000, 391), (1999, 474), (129, 33), (64, 8), (3, 2), (2500, 223)):
    X = rng.standard_normal((n, p)) * np.exp(rng.standard_normal((n, 1)))
    X[: n // 3] += 3 * rng.standard_normal((1, p))
    Xc = X - X.mean(1, keepdims=True)
    U = np.ascontiguousarray(Xc / np.sqrt((Xc * Xc).sum(1, keepdims=True)))
    D = np.zeros((n, n))
    rc = lib.sharp_dist_i8(dp(U), n, p, dp(D))
    assert rc == 0, lib.sharp_last_error()
    Ul = U.astype(np.longdouble)
    ref = 1 - np.clip(Ul @ Ul.T, -1, 1); np.fill_diagonal(ref, 0)
    e8 = np.abs(D - ref.astype(np.float64)).max()
    # the fp64 MFMA kernel on the same rows (operands k-major)
    At = np.ascontiguousarray(U.T)
    D64 = np.zeros((n, n))
    rc = lib.sharp_gemm_tn_f64(dp(At), dp(At), dp(D64), n, n, p, 1, 1, 1)
    assert rc == 0, lib.sharp_last_error()
    e64 = np.abs(D64 - ref.astype(np.float64)).max()
    print("n=%d p=%d: max |D_i8 - ref| = %.3e   max |D_f64mfma - ref| = %.3e   symmetric=%s diag0=%s" % (n, p, e8, e64, np.array_equal(D, D.T), not D.diagonal().any()))
