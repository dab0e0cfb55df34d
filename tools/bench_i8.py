"""usage: bench_i8.py [n p count reps]: the digit kernel, the integer product kernel and the fp64 MFMA kernel on `count` tasks."""
import sys, ctypes as C
import numpy as np
sys.path.insert(0, "/root/repo")
import os
import sharp_amd
from sharp_amd import _lib as _L
if os.environ.get('SHARP_VARIANT'):
    _L._SO = os.path.join(os.path.dirname(_L._SO), 'variants', 'libsharp_hip_%s.so' % os.environ['SHARP_VARIANT'])
sharp_amd.init(0); lib = sharp_amd.lib()
a = [int(v) for v in sys.argv[1:]]
n, p, count, reps = (a + [2000, 391, 188, 5][len(a):])[:4]
ms = (C.c_double * 3)()
rc = lib.sharp_dist_i8_bench(n, p, count, reps, ms)
assert rc == 0, lib.sharp_last_error()
print("n=%d p=%d tasks=%d: digits %.3f ms, products %.3f ms, fp64 MFMA %.3f ms" % (n, p, count, ms[0], ms[1], ms[2]))
