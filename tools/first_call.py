import sys, time, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import sharp_amd
from sharp_amd import device as dev
sharp_amd.init(0)
n, m = 50000, 20000
dX = torch.empty((n, m), dtype=torch.float32, device="cuda")
dev.synth_fill(dX, 20261003, 0, 12, 1000)
torch.cuda.synchronize()
for i in range(3):
    dev.profile(True)
    t0 = time.perf_counter()
    dev.SHARP_dev(dX, ensize_K=15, rN_seed=2103)
    dt = time.perf_counter() - t0
    tab = dev.profile_table()
    print("call %d: %.1f ms  alloc %.1f ms  proj %.1f  alloc_E %.1f" % (i, dt * 1e3, tab.get("host:hc_workspace_alloc", (0, 0))[0], tab.get("host:projector_build", (0, 0))[0], tab.get("host:alloc_E", (0,0))[0]))
