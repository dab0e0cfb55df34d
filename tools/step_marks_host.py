"""SHARP_STEP_MARKS=1 of SHARP_unlimited on a LIST of sparse HOST blocks (cfg3 shape): where a block's upload spends its time beside the clustering."""
import os, sys
os.environ["SHARP_STEP_MARKS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
import sharp_amd
from sharp_amd import device as dev
sharp_amd.init(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 6
nb, m = 50000, 20000
blocks = []
for b in range(B):
    x = torch.empty((nb, m), dtype=torch.float32, device="cuda"); dev.synth_fill(x, 20261003, b * nb, 12, 1000)
    nz = x.nonzero()
    cp = np.concatenate([[0], np.cumsum(torch.bincount(nz[:, 0], minlength=nb).cpu().numpy())]).astype(np.int32)
    blocks.append(sp.csc_matrix((x[nz[:, 0], nz[:, 1]].double().cpu().numpy(), nz[:, 1].int().cpu().numpy(), cp), shape=(m, nb)))
    del x, nz
for i in range(3):
    sys.stderr.write("==== call %d\n" % i); sys.stderr.flush()
    sharp_amd.SHARP_unlimited(blocks, ensize_K=5, rN_seed=2103, viewflag=False)
