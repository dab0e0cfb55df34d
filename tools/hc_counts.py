"""How many base-clustering tasks took the bulk-synchronous / sequential agglomeration in one SHARP() call (cfg2 shape)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sharp_amd
from sharp_amd import device as dev
sharp_amd.init(0)
n, m = 50000, 20000
dX = torch.empty((n, m), dtype=torch.float32, device="cuda")
dev.synth_fill(dX, 20261003, 0, 12, 1000)
for K in (int(a) for a in (sys.argv[1:] or ["15"])):
    dev.SHARP_dev(dX, ensize_K=K, rN_seed=2103)
    dev.profile(True)
    dev.SHARP_dev(dX, ensize_K=K, rN_seed=2103)
    t = dev.profile_table()
    print(K, {k: v for k, v in t.items() if "hclust" in k or "corr_dist" in k})
