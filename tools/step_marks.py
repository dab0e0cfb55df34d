"""The host-side milestones of one cfg3 call (SHARP_STEP_MARKS=1): ten resident blocks, the third call's marks on stderr."""
import os, sys, ctypes as C
os.environ["SHARP_STEP_MARKS"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sharp_amd
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
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    sys.stderr.write("==== call %d\n" % i); sys.stderr.flush()
    call()
