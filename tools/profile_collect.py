"""python tools/profile_collect.py r01 : gpurun_out/<tag>_{kt,pmc_fetch,pmc_write} -> profiles/<tag>_*.csv, rp_traffic.json.

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE are reported in KB by rocprofv3;
on gfx950 FETCH_SIZE counts half of the bytes of wide coalesced reads, so it is doubled (an upper bound for the
narrow-access kernels); WRITE_SIZE is taken as is.  Collected in separate passes."""
import csv
import glob
import json
import os
import re
import shutil
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
# round 5 on: the profiled command is the default bench (cfg3: 1 warm-up + 3 timed SHARP_unlimited calls = 4 calls of 10 blocks), or cfg2
# (1 warm-up + 3 timed + 1 attribution step = 5 SHARP() calls): `profile_collect.py TAG CALLS SHAPE`
CALLS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
SHAPE = sys.argv[3] if len(sys.argv) > 3 else "cfg2"
LAUNCHES_PER_CALL = {"cfg2": 1, "cfg3": 10, "cfg4": 8}[SHAPE]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out")
prof = os.path.join(root, "profiles")
os.makedirs(prof, exist_ok=True)


def newest(pattern):
    files = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)      # the latest run
    return files[-1] if files else None


def short(name):
    name = re.sub(r"\(.*$", "", name)
    return name.strip().strip('"')


ks = newest(os.path.join(out, tag + "_kt", "**", "*kernel_stats.csv"))
if ks:
    shutil.copy(ks, os.path.join(prof, tag + "_bench_kernel_stats.csv"))
    print("kernel stats:", ks)

agg = defaultdict(lambda: {"launches": 0, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})
for counter, sub in (("FETCH_SIZE", "_pmc_fetch"), ("WRITE_SIZE", "_pmc_write")):
    f = newest(os.path.join(out, tag + sub, "**", "*counter_collection.csv"))
    if not f:
        continue
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if row["Counter_Name"] != counter:
                continue
            k = short(row["Kernel_Name"])
            agg[k][counter] += float(row["Counter_Value"])
            if counter == "FETCH_SIZE":
                agg[k]["launches"] += 1
steps = CALLS
rows = []
for k, v in sorted(agg.items(), key=lambda kv: -(2 * kv[1]["FETCH_SIZE"] + kv[1]["WRITE_SIZE"])):
    hbm = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
    rows.append((k, v["launches"], round(v["FETCH_SIZE"], 1), round(v["WRITE_SIZE"], 1), int(hbm), int(hbm / steps)))
with open(os.path.join(prof, tag + "_pmc_hbm_traffic.csv"), "w") as fh:
    fh.write("kernel,launches(%d calls of the %s workload),FETCH_SIZE_KB_sum,WRITE_SIZE_KB_sum,"
             "hbm_bytes_total(2xFETCH gfx950 correction + WRITE),hbm_bytes_per_call\n" % (steps, SHAPE))
    for r in rows:
        fh.write(",".join(str(x) for x in r) + "\n")
rp = [r for r in rows if "rp_compact_kernel" in r[0] or "rp_apply_kernel" in r[0] or "rp_pc_kernel" in r[0]]
if rp:
    per_launch = sum(r[4] for r in rp) / max(1, sum(r[1] for r in rp if "rp_pc_kernel" in r[0] or "rp_compact_kernel" in r[0]))
    tf = os.path.join(prof, "rp_traffic.json")
    tj = json.load(open(tf)) if os.path.exists(tf) else {}
    tj[SHAPE] = {"hbm_bytes_per_launch": int(per_launch), "kernel": rp[0][0], "source": "profiles/%s_pmc_hbm_traffic.csv (rocprofv3 --pmc passes over bench.py --config %s)" % (tag, SHAPE)}
    json.dump(tj, open(tf, "w"), indent=1)
    print("rp traffic per launch (%s): %.3f GB" % (SHAPE, per_launch / 1e9))
b = os.path.join(out, tag + "_bench_n1.json")
if os.path.exists(b) and os.path.getsize(b) > 10:
    shutil.copy(b, os.path.join(prof, tag + "_bench_n1.json"))
    print("bench:", open(b).read()[:200])
src = os.path.join(out, tag + "_bench_under_rocprof.json")
if os.path.exists(src) and os.path.getsize(src) > 0:
    shutil.copy(src, os.path.join(prof, tag + "_bench_under_rocprof.json"))
