"""Scratch micro-benchmark of the RP scatter kernel alone (cfg2/cfg3 shapes)."""
import ctypes as C
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import os
import sharp_amd
from sharp_amd import _lib as _L
if os.environ.get('SHARP_VARIANT'):
    _L._SO = os.path.join(os.path.dirname(_L._SO), 'variants', 'libsharp_hip_%s.so' % os.environ['SHARP_VARIANT'])

sharp_amd.init(0)
lib = sharp_amd.lib()
SEED = 20261003
CFGS = [(20000, 50000, 15, 391), (20000, 50000, 5, 474), (27000, 40000, 5, 508), (20000, 50000, 15, 466), (20000, 50000, 15, 441),   # 3, 4: a rank of an 8- / 4-GPU run
        (27000, 162500, 5, 508)]   # 5: cfg4's per-GPU share at full size
if len(sys.argv) > 1:
    CFGS = [CFGS[int(a)] for a in sys.argv[1:]]
for (m, n, K, p) in CFGS:
    t0 = time.time()
    pr = sharp_amd.Projector(m, p, [50 + 2103 + k for k in range(1, K + 1)])
    tproj = time.time() - t0
    dX = torch.empty((n, m), dtype=torch.float32, device="cuda")
    lib.sharp_synth_fill_dev(C.c_uint(SEED), m, C.c_longlong(0), n, 12, 1000, C.c_void_p(dX.data_ptr()), C.c_longlong(m))
    lib.sharp_synchronize()
    dE = torch.zeros((n, K * p), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    lib.sharp_profile_enable(1)
    for it in range(3):
        lib.sharp_project_dev(pr.handle, C.c_void_p(dX.data_ptr()), m, n, C.c_longlong(m), 1, C.c_void_p(dE.data_ptr()),
                              C.c_longlong(K * p))
    lib.sharp_synchronize()
    lib.sharp_profile_reset()
    reps = 10
    for it in range(reps):
        lib.sharp_project_dev(pr.handle, C.c_void_p(dX.data_ptr()), m, n, C.c_longlong(m), 1, C.c_void_p(dE.data_ptr()),
                              C.c_longlong(K * p))
    lib.sharp_synchronize()
    ms = C.c_double()
    cnt = C.c_longlong()
    lib.sharp_profile_get(b"rp_stage", C.byref(ms), C.byref(cnt))
    t = ms.value / cnt.value * 1e-3
    rd = n * m * 4
    wr = n * K * p * 8
    nzfrac = float((dX[:2000] != 0).float().mean())
    per = []
    for nm in (b"rp_compact", b"rp_apply", b"rp_pc"):
        lib.sharp_profile_get(nm, C.byref(ms), C.byref(cnt))
        per.append("%s %.1f us x%d" % (nm.decode(), ms.value / max(cnt.value, 1) * 1e3, cnt.value // reps))
    print(f"m={m} n={n} K={K} p={p} nnz={pr.nnz()} proj_build={tproj:.2f}s  rp={t*1e3:.3f} ms  read {rd/t/1e12:.2f} TB/s "
          f"({rd/t/8e12*100:.1f}% of 8TB/s)  read+write {(rd+wr)/t/1e12:.2f} TB/s  nz={nzfrac:.3f}  " + "  ".join(per), flush=True)
    del dX, dE
