import sys, time, ctypes as C
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import os
import sharp_amd
from sharp_amd import _lib as _L
if os.environ.get('SHARP_VARIANT'):
    _L._SO = os.path.join(os.path.dirname(_L._SO), 'variants', 'libsharp_hip_%s.so' % os.environ['SHARP_VARIANT'])
from sharp_amd import device as dev
sharp_amd.init(0); lib = sharp_amd.lib()
B, nb, m, K = 10, 50000, 20000, 5
blocks = []
for b in range(B):
    x = torch.empty((nb, m), dtype=torch.float32, device="cuda"); dev.synth_fill(x, 20261003, b * nb, 12, 1000); blocks.append(x)
def call():
    ptrs = (C.c_void_p * B)(*[b.data_ptr() for b in blocks])
    ncb = np.array([nb] * B, np.int64); ldb = np.array([m] * B, np.int64)
    pred = np.zeros(B * nb, np.int32); npred, pu = C.c_int(), C.c_int()
    rc = lib.sharp_SHARP_unlimited_dev(ptrs, ncb.ctypes.data_as(C.POINTER(C.c_longlong)), ldb.ctypes.data_as(C.POINTER(C.c_longlong)), B, m, K, 0, 0, 0, C.c_double(2103), pred.ctypes.data_as(C.POINTER(C.c_int)), C.byref(npred), C.byref(pu))
    assert rc in (0, 16, 32, 48), lib.sharp_last_error()
    return pred
call()
dev.profile(True)
t0 = time.perf_counter(); call(); torch.cuda.synchronize(); t1 = time.perf_counter()
prof = dev.profile_table()
print("call %.1f ms" % ((t1 - t0) * 1e3))
for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:22]:
    print("%-34s %8.2f ms  x%d" % (k, v[0], v[1]))
