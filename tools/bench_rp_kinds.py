"""The RP matmul stage alone (sharp_project_dev / _dev64) on one block of a shape, for the kinds of values the reference meets
(R/SHARP.R:110-114,343-345,569-571): fp32-exact counts (the term table), counts with a UMI-like heavy tail, CPM-normalised counts and
TPM-like doubles (fp64 blocks).  usage: python tools/bench_rp_kinds.py [cfg3|cfg2|cfg4] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench

shape = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
Bn = bench.Bench(np, torch, 0)
n, m, K, p = bench.SHAPES[shape]
x = Bn.synth_block(0, n, m)
g = torch.Generator(device="cuda"); g.manual_seed(7)


def show(tag, r):
    print("%-28s %7.3f ms  read %6.1f GB/s at %d B stored = %.3f of 8 TB/s   (read+write %.3f)" % (tag, r["ms"], r["achieved_read"], r["stored_width_bytes"], r["frac_read"], r["frac_read_write"]), flush=True)


print("%s block: %d cells x %d genes, K = %d, p = %d; non-zeros %.1f %%" % (shape, n, m, K, p, 100.0 * float((x != 0).float().mean())))
show("counts (table path)", Bn.rp_stage_alone(x, K, p, reps))
xt = x.clone()
hit = (torch.rand(xt.shape, device="cuda", generator=g) < 1e-3) & (xt != 0)
xt[hit] = torch.randint(256, 4096, (int(hit.sum().item()),), device="cuda", generator=g).float()
show("counts, 0.1 % heavy tail", Bn.rp_stage_alone(xt, K, p, reps))
del xt, hit
xd = x.double()
xd = xd / xd.sum(1, keepdim=True).clamp_min(1.0) * 1e6
show("CPM doubles (fp64 block)", Bn.rp_stage_alone(xd, K, p, reps))
gl = 0.5 + torch.rand((1, m), device="cuda", generator=g, dtype=torch.float64) * 4.0
xd = x.double() / gl
xd = xd / xd.sum(1, keepdim=True).clamp_min(1e-300) * 1e6
show("TPM-like doubles (fp64 block)", Bn.rp_stage_alone(xd, K, p, reps))
# a denser TPM-like block: 30 % non-zero (full-length protocols detect 5-8 thousand genes per cell)
dense = (torch.rand(x.shape, device="cuda", generator=g) < 0.22) & (x == 0)
xd = x.double()
xd[dense] = torch.rand((int(dense.sum().item()),), device="cuda", generator=g, dtype=torch.float64) * 3.0 + 0.01
del dense
xd = xd / xd.sum(1, keepdim=True).clamp_min(1e-300) * 1e6
print("dense TPM-like block: non-zeros %.1f %%" % (100.0 * float((xd != 0).float().mean())))
show("TPM-like, 30 % non-zero", Bn.rp_stage_alone(xd, K, p, reps))
os.environ["SHARP_RP_KERNEL"] = "split"
Bn.sa.reload_options()
show("  same, two-kernel form", Bn.rp_stage_alone(xd, K, p, reps))
