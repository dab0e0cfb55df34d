"""fp64 blocks through the RP stage: kernel form (pc / split) x accumulator mode (dual / signed codes), two densities.  Lab tool."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench

shape = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
Bn = bench.Bench(np, torch, 0)
n, m, K, p = bench.SHAPES[shape]
x = Bn.synth_block(0, n, m)
g = torch.Generator(device="cuda"); g.manual_seed(7)
xd = x.double()
cpm = xd / xd.sum(1, keepdim=True).clamp_min(1.0) * 1e6
dense = (torch.rand(x.shape, device="cuda", generator=g) < 0.22) & (x == 0)
xd[dense] = torch.rand((int(dense.sum().item()),), device="cuda", generator=g, dtype=torch.float64) * 3.0 + 0.01
del dense
tpm30 = xd / xd.sum(1, keepdim=True).clamp_min(1e-300) * 1e6
del xd
for env in ({}, {"SHARP_RP_DUAL": "0"}, {"SHARP_RP_KERNEL": "split"}, {"SHARP_RP_KERNEL": "split", "SHARP_RP_DUAL": "0"}, {"SHARP_RP_PC_WGS": "1"}):
    for k in ("SHARP_RP_DUAL", "SHARP_RP_KERNEL", "SHARP_RP_PC_WGS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    Bn.sa.reload_options()
    r0 = Bn.rp_stage_alone(x, K, p, 6)
    r1 = Bn.rp_stage_alone(cpm, K, p, 6)
    r2 = Bn.rp_stage_alone(tpm30, K, p, 6)
    print("%-50s counts fp32 %.3f ms | CPM fp64 (11 %% nz) %.3f ms | TPM-like fp64 (31 %% nz) %.3f ms" % (env or "default (pc, dual)", r0["ms"], r1["ms"], r2["ms"]), flush=True)
