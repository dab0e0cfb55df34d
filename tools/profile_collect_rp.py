"""python tools/profile_collect_rp.py r03: gpurun_out/<tag>_rpk5_* -> profiles/<tag>_rp_k5_kernel_stats.csv (the two shapes' kernel
statistics one after the other) and profiles/<tag>_rp_k5_traffic.txt (HBM bytes per RP stage from FETCH_SIZE / WRITE_SIZE, collected in
separate passes; FETCH_SIZE doubled as /opt/skills/guides/MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out, prof = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
shapes = {1: ("cfg3_block", 50000, 20000, 5, 474), 5: ("cfg4_share", 162500, 27000, 5, 508)}
CALLS = 13                                     # bench_rp.py: 3 warm-up + 10 timed stage runs per process


def newest(pattern):
    files = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return files[-1] if files else None


with open(os.path.join(prof, tag + "_rp_k5_kernel_stats.csv"), "w") as fo:
    for cfg, (name, n, m, K, p) in shapes.items():
        f = newest(os.path.join(out, "%s_rpk5_kt_%d" % (tag, cfg), "**", "*kernel_stats.csv"))
        if not f:
            continue
        fo.write("# %s: RP stage alone, %d cells x %d genes, K = %d, p = %d, %d stage runs (tools/bench_rp.py %d under rocprofv3 --kernel-trace --stats)\n"
                 % (name, n, m, K, p, CALLS, cfg))
        log = os.path.join(out, "%s_rpk5_kt_%d.log" % (tag, cfg))
        if os.path.exists(log):
            for ln in open(log):
                if ln.startswith("m="):
                    fo.write("# bench_rp line of the same process (HIP events): " + ln)
        for ln in open(f):
            if "rp_" in ln or ln.startswith('"Name"') or ln.startswith("Name"):
                fo.write(ln)
with open(os.path.join(prof, tag + "_rp_k5_traffic.txt"), "w") as fo:
    for cfg, (name, n, m, K, p) in shapes.items():
        agg = defaultdict(lambda: defaultdict(float))
        for counter, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
            f = newest(os.path.join(out, "%s_rpk5_%s_%d" % (tag, sub, cfg), "**", "*counter_collection.csv"))
            if not f:
                continue
            for row in csv.DictReader(open(f)):
                k = re.sub(r"\(.*$", "", row["Kernel_Name"]).strip()
                if "rp_" in k and row["Counter_Name"] == counter:
                    agg[k][counter] += float(row["Counter_Value"])
        rd = sum(2 * v["FETCH_SIZE"] * 1024 for v in agg.values()) / CALLS
        wr = sum(v["WRITE_SIZE"] * 1024 for v in agg.values()) / CALLS
        alg = n * m * 4
        fo.write("%s (%d x %d, K = %d, p = %d): per RP stage HBM read %.2f GB (FETCH_SIZE x 2), write %.2f GB, total %.2f GB; algorithmic read %.2f GB "
                 "(X once for all K), E written %.2f GB fp64 -> traffic / algorithmic read = %.2f\n"
                 % (name, n, m, K, p, rd / 1e9, wr / 1e9, (rd + wr) / 1e9, alg / 1e9, n * K * p * 8 / 1e9, (rd + wr) / alg))
        for k, v in sorted(agg.items()):
            fo.write("    %-60s read %.3f GB  write %.3f GB per stage\n" % (k[:60], 2 * v["FETCH_SIZE"] * 1024 / CALLS / 1e9, v["WRITE_SIZE"] * 1024 / CALLS / 1e9))
print(open(os.path.join(prof, tag + "_rp_k5_traffic.txt")).read())
