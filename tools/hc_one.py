"""One clustering task of n observations through sharp_get_opt_hclust (debug aid for the agglomeration kernels)."""
import sys
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import os
import sharp_amd
from sharp_amd import _lib as _L
if os.environ.get('SHARP_VARIANT'):
    _L._SO = os.path.join(os.path.dirname(_L._SO), 'variants', 'libsharp_hip_%s.so' % os.environ['SHARP_VARIANT'])
sharp_amd.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
hm = sys.argv[2] if len(sys.argv) > 2 else "ward.D"
rng = np.random.default_rng(11)
E = rng.standard_normal((n, 60)) + np.repeat(rng.standard_normal((7, 60)) * 2.0, (n + 6) // 7, axis=0)[:n]
print("start", n, hm, flush=True)
a = sa = sharp_amd.get_opt_hclust(E, hmethod=hm)
os.environ["SHARP_HC_SEQ"] = "1"
b = sharp_amd.get_opt_hclust(E, hmethod=hm)
print("labels equal", np.array_equal(a["v"], b["v"]), "max rel height diff", np.max(np.abs(a["height"] - b["height"]) / np.abs(b["height"])), flush=True)
