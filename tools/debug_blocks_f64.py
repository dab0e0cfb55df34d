import os, sys, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sharp_amd
from sharp_amd import blocks as B, device as dev
from oracle import pyoracle as orc
orc.build(); sharp_amd.init(0)
m = 1500; sizes = [5200, 700, 5300]
offs = np.concatenate([[0], np.cumsum(sizes)])
parts = []
for i, nb in enumerate(sizes):
    X = orc.synth_fill(77, m, int(offs[i]), nb, 5, 150)
    parts.append(X / np.maximum(X.sum(0, keepdims=True), 1.0) * 1e6)
d = os.path.join(tempfile.gettempdir(), "sharpblk_dbgxyz"); shutil.rmtree(d, ignore_errors=True); os.mkdir(d)
for i, X in enumerate(parts):
    B.write_block(os.path.join(d, "p%d.blk" % (i + 1)), X)
ref = orc.SHARP_unlimited(parts, K=3, rN_seed=7, nthreads=8)
host = sharp_amd.SHARP_unlimited(parts, ensize_K=3, rN_seed=7, viewflag=False)
print("host list vs oracle:", np.array_equal(host["pred_clusters"], ref["pred_clusters"]))
nd = {"dir": d, "ncells": int(offs[-1]), "ngenes": m}
a = sharp_amd.SHARP_unlimited3(nd, ensize_K=3, rN_seed=7, viewflag=True)
print("unlimited3 viewflag (block by block) vs oracle:", np.array_equal(a["pred_clusters"], ref["pred_clusters"]))
b = sharp_amd.SHARP_unlimited3(nd, ensize_K=3, rN_seed=7, viewflag=False)
print("unlimited3 grouped vs oracle:", np.array_equal(b["pred_clusters"], ref["pred_clusters"]))
# the expanded blocks themselves
st = B.BlockStreamer(B.list_block_files(d))
for i, h, x in st:
    print("block", i, h["val_bits"], x.dtype, tuple(x.shape), x.stride(), "equal to the source:", np.array_equal(x.cpu().numpy(), parts[i].T))
# grouped through device entry directly on resident fp64 tensors
ts = [torch.from_numpy(np.ascontiguousarray(p.T)).cuda() for p in parts]
p = int(np.ceil(np.log2(int(offs[-1])) / 0.04))
proj = sharp_amd.Projector(m, p, [50 + 7 + k for k in range(1, 4)])
res = dev.unlimited_blocks_dev(ts, p, proj.handle, 3, 7)
one = [dev.unlimited_block_dev(t, p, proj.handle, 3, 7) for t in ts]
for q in range(3):
    print("block", q, "blocks_dev == block_dev labels:", np.array_equal(res[q][0], one[q][0]), "clusters", res[q][1].shape[0], one[q][1].shape[0])
shutil.rmtree(d, ignore_errors=True)
