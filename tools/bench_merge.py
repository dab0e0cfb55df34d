"""Times sharp_unlimited_merge at the shapes of the large configurations (synthetic centroid tables: B blocks x ~11 block-level clusters
around 12 planted centres): cfg4 (8 blocks, 1.3 M cells: k = 26 .. nC - 1), cfg5 (200 blocks, 1e7 cells: k = 200 .. 2000).
Prints the wall time and the per-kernel table."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import sharp_amd
from sharp_amd import device as dev

sharp_amd.init(0)
rng = np.random.default_rng(1)
for tag, B, per, p, ncells in [("cfg4", 8, 11, 508, 1300000), ("cfg5", 200, 11, 582, 10000000), ("cfg5-dense", 200, 40, 582, 10000000)]:
    cen = rng.standard_normal((12, p)) * 2
    M = cen[rng.integers(0, 12, B * per)] + 0.4 * rng.standard_normal((B * per, p))
    Cn = rng.integers(200, 8000, B * per).astype(np.int64)
    for rep in range(2):
        dev.profile(True)
        t0 = time.perf_counter()
        fid, nf = dev.unlimited_merge(M, Cn, ncells)
        dt = time.perf_counter() - t0
        prof = dev.profile_table()
    top = {k: round(v[0], 2) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:10]}
    print(json.dumps({"case": tag, "rows": B * per, "final": nf, "seconds": round(dt, 4), "top_ms": top}), flush=True)
