import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sharp_amd
from sharp_amd import _lib as _L
if os.environ.get('SHARP_VARIANT'):
    _L._SO = os.path.join(os.path.dirname(_L._SO), 'variants', 'libsharp_hip_%s.so' % os.environ['SHARP_VARIANT'])
sharp_amd.init(0)
from sharp_amd import device as dev
rng = np.random.default_rng(3)
n = int(sys.argv[1])
centers = rng.standard_normal((6, 30)) * 3
X = centers[rng.integers(0, 6, n)] + rng.standard_normal((n, 30))
dev.profile(True)
print("go", flush=True)
t0 = time.time()
r = sharp_amd.get_opt_hclust(X)
print("done %.3f s k=%d" % (time.time() - t0, r["optN_cluster"]), flush=True)
print({k: v for k, v in dev.profile_table().items() if "hclust" in k}, flush=True)
