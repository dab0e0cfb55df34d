"""Where the forview / viewflag outputs' extra time goes: host-timer table of a labels-only step and of a view step, cfg2 (SHARP()) and cfg3
(SHARP_unlimited).  usage: python tools/forview_profile.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench

Bn = bench.Bench(np, torch, 0)
dev = Bn.dev


def run(tag, fn, reps=4):
    fn(); fn()
    dev.profile(True)
    Bn.lib.sharp_synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    Bn.lib.sharp_synchronize()
    dt = (time.perf_counter() - t0) / reps
    tab = dev.profile_table()
    dev.profile(False)
    print("%-28s %.2f ms per call" % (tag, dt * 1e3))
    return dt, {k: v[0] / reps for k, v in tab.items()}


def diff(a, b, top=14):
    keys = sorted(set(a) | set(b), key=lambda k: -(b.get(k, 0) - a.get(k, 0)))
    for k in keys[:top]:
        print("     %-34s %8.3f -> %8.3f ms" % (k, a.get(k, 0), b.get(k, 0)))


x = Bn.synth_block(0, 50000, 20000)
_, a = run("cfg2 labels only", lambda: dev.SHARP_dev(x, ensize_K=15, rN_seed=2103))
keep = {}


def view_call():
    r = dev.SHARP_dev(x, ensize_K=15, rN_seed=2103, forview=True, view_out=keep.get("b"))
    keep["b"] = r[1]["view_out"]


_, b0 = run("cfg2 forview, fresh buffers", lambda: dev.SHARP_dev(x, ensize_K=15, rN_seed=2103, forview=True))
_, b = run("cfg2 forview, kept buffers", view_call)
diff(a, b)
del x
blocks = [Bn.synth_block(b * 50000, 50000, 20000) for b in range(10)]
_, a = run("cfg3 labels only", lambda: Bn.unlimited_call(blocks, 5))
_, b = run("cfg3 viewflag", lambda: Bn.unlimited_call(blocks, 5, view=True))
diff(a, b)
t0 = time.perf_counter()
for _ in range(5):
    from sharp_amd.api import _one_hot
    _one_hot(np.ones(500000, np.int32), 10)
print("one-hot x0 (500 000 cells): %.2f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
