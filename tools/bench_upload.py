"""Where a sparse host block's way into HBM spends its time (sharp_csc_to_dense_dev on one cfg3 block of counts), by packing threads."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sharp_amd
from sharp_amd import device as dev
sharp_amd.init(0); lib = sharp_amd.lib()
nb, m = 50000, 20000
x = torch.empty((nb, m), dtype=torch.float32, device="cuda"); dev.synth_fill(x, 20261003, 0, 12, 1000)
nz = x.nonzero()
cp = np.concatenate([[0], np.cumsum(torch.bincount(nz[:, 0], minlength=nb).cpu().numpy())]).astype(np.int32)
ri = nz[:, 1].int().cpu().numpy(); xv = x[nz[:, 0], nz[:, 1]].double().cpu().numpy()
del nz
dX = torch.zeros((nb, m), dtype=torch.float32, device="cuda")
ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int)); dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
print("nnz %.3g: R holds %.2f GB, wire %.2f GB" % (xv.size, xv.size * 12 / 1e9, xv.size * 3 / 1e9))
for thr in ("", "8", "16", "32", "48", "96"):
    if thr: os.environ["SHARP_UPLOAD_THREADS"] = thr
    else: os.environ.pop("SHARP_UPLOAD_THREADS", None)
    sharp_amd.reload_options()
    ts = []
    for it in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rc = lib.sharp_csc_to_dense_dev(ip(cp), ip(ri), dp(xv), m, C.c_longlong(nb), C.c_void_p(dX.data_ptr()), C.c_longlong(m))
        assert rc == 0, lib.sharp_last_error()
        ts.append((time.perf_counter() - t0) * 1e3)
    print("SHARP_UPLOAD_THREADS=%-3s  %s ms per block; equal to the source: %s" % (thr or "dflt", " ".join("%.1f" % t for t in ts), bool(torch.equal(dX, x))))
dev.profile(True)
lib.sharp_csc_to_dense_dev(ip(cp), ip(ri), dp(xv), m, C.c_longlong(nb), C.c_void_p(dX.data_ptr()), C.c_longlong(m))
for k, v in sorted(dev.profile_table().items(), key=lambda kv: -kv[1][0])[:12]:
    print("   %-30s %8.2f ms x%d" % (k, v[0], v[1]))
