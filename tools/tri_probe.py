"""Scratch: the upper-triangle agglomeration kernel on one small task, against the full-matrix kernel (SHARP_HC_TRI=0)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sharp_amd

sharp_amd.init(0)
rng = np.random.default_rng(3)
for n in [int(a) for a in sys.argv[1:]] or [40, 300, 900]:
    centers = rng.standard_normal((6, 30)) * 3
    X = centers[rng.integers(0, 6, n)] + rng.standard_normal((n, 30))
    out = {}
    for tri in ("1", "0"):
        os.environ["SHARP_HC_TRI"] = tri
        sharp_amd.reload_options()
        print("n=%d tri=%s ..." % (n, tri), flush=True)
        t0 = time.time()
        r = sharp_amd.get_opt_hclust(X)
        print("   done in %.3f s, k=%d maxsil=%.6f" % (time.time() - t0, r["optN_cluster"], r["maxsil"]), flush=True)
        out[tri] = r
    same = np.array_equal(out["1"]["v"], out["0"]["v"]) and np.allclose(out["1"]["height"], out["0"]["height"], rtol=1e-12, atol=0)
    print("n=%d: every cutree level equal and heights equal: %s" % (n, same), flush=True)
