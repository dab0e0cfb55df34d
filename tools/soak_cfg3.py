"""cfg3 soak: N calls of SHARP_unlimited on ten resident blocks back to back after two warm-up calls; every call's labels against the first call's,
call times, free device memory and the process's thread count before and after (usage: soak_cfg3.py [N=40])."""
import os, sys, time, ctypes as C
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
def threads():
    return len(os.listdir("/proc/self/task"))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ref = call(); call()
free0 = torch.cuda.mem_get_info()[0]; th0 = threads()
ts, same = [], 0
for i in range(N):
    torch.cuda.synchronize(); t0 = time.perf_counter(); p = call(); ts.append((time.perf_counter() - t0) * 1e3)
    same += int(np.array_equal(p, ref))
free1 = torch.cuda.mem_get_info()[0]
ts.sort()
print("%d calls: labels identical to the first call's in %d; min %.1f median %.1f max %.1f ms; free memory %.2f -> %.2f GB; process threads %d -> %d"
      % (N, same, ts[0], ts[len(ts) // 2], ts[-1], free0 / 1e9, free1 / 1e9, th0, threads()))
