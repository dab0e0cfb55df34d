"""The RP matmul stage alone on one cfg3 block, for counter collection: `rp_one.py f32|f64 [reps]` (f64: CPM-normalised doubles)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench

kind = sys.argv[1] if len(sys.argv) > 1 else "f32"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
Bn = bench.Bench(np, torch, 0)
n, m, K, p = bench.SHAPES["cfg3"]
x = Bn.synth_block(0, n, m)
if kind == "f64":
    x = x.double()
    x = x / x.sum(1, keepdim=True).clamp_min(1.0) * 1e6
r = Bn.rp_stage_alone(x, K, p, reps)
print("%s: %.3f ms, read %.1f GB/s at %d B stored = %.3f of 8 TB/s" % (kind, r["ms"], r["achieved_read"], r["stored_width_bytes"], r["frac_read"]))
