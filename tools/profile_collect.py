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
steps = 5  # bench.py --steps 3 --warmup 1: 1 warm-up + 3 timed + 1 attribution step = 5 SHARP() calls
rows = []
for k, v in sorted(agg.items(), key=lambda kv: -(2 * kv[1]["FETCH_SIZE"] + kv[1]["WRITE_SIZE"])):
    hbm = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
    rows.append((k, v["launches"], round(v["FETCH_SIZE"], 1), round(v["WRITE_SIZE"], 1), int(hbm), int(hbm / steps)))
with open(os.path.join(prof, tag + "_pmc_hbm_traffic.csv"), "w") as fh:
    fh.write("kernel,launches(%d SHARP calls: 1 warm-up + 3 timed + 1 attribution),FETCH_SIZE_KB_sum,WRITE_SIZE_KB_sum,"
             "hbm_bytes_total(2xFETCH gfx950 correction + WRITE),hbm_bytes_per_SHARP_call\n" % steps)
    for r in rows:
        fh.write(",".join(str(x) for x in r) + "\n")
rp = [r for r in rows if "rp_compact_kernel" in r[0] or "rp_apply_kernel" in r[0] or "rp_pc_kernel" in r[0]]
if rp:
    per_call = sum(r[5] for r in rp)
    launches = {r[0]: r[1] // steps for r in rp}
    json.dump({"kernels": [r[0] for r in rp], "launches_per_SHARP_call": launches, "hbm_bytes_per_launch": per_call,
               "hbm_bytes_per_kernel_per_SHARP_call": {r[0]: r[5] for r in rp},
               "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `python3 bench.py --steps 3 --warmup 1 "
                         "--no-cpu-baseline` (tools/profile_round.sh); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports "
                         "half of wide coalesced reads); summed over the RP stage's launches of one SHARP() call (rp_pc_kernel: one launch; the two-kernel form: compact + apply launches)",
               "workload": "bench.py default (50000 cells x 20000 genes, K=15, p=391)"},
              open(os.path.join(prof, "rp_traffic.json"), "w"), indent=1)
    print("rp traffic per SHARP call: %.3f GB" % (per_call / 1e9))
b = os.path.join(out, tag + "_bench_n1.json")
if os.path.exists(b) and os.path.getsize(b) > 10:
    shutil.copy(b, os.path.join(prof, tag + "_bench_n1.json"))
    print("bench:", open(b).read()[:200])
src = os.path.join(out, tag + "_bench_under_rocprof.json")
if os.path.exists(src) and os.path.getsize(src) > 0:
    shutil.copy(src, os.path.join(prof, tag + "_bench_under_rocprof.json"))
