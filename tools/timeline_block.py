"""Kernel timeline of the LAST sharp_unlimited_block_dev call of a `rocprofv3 --kernel-trace` run of tools/block_profile.py:
the call starts at the last rp_fixtab / first rp_compact_kernel after a gap; prints kernels longer than min_us and the idle gaps."""
import csv, glob, os, sys
root = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
files = sorted(glob.glob(root + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = []
for r in csv.DictReader(open(files[-1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if "rp_compact_kernel" in r[3] and (i == 0 or "rp_" not in rows[i - 1][3])]
i0 = starts[-1]
t0 = rows[i0][0]
last_end = t0
for s, e, q, name in rows[i0:]:
    short = name.split("(")[0].replace("void ", "").replace("sharp::", "").replace("(anonymous namespace)::", "")
    if s - last_end > 200e3:
        print("%8.3f           idle %7.3f ms" % ((last_end - t0) / 1e6, (s - last_end) / 1e6))
    if (e - s) / 1e3 >= min_us:
        print("%8.3f -> %8.3f  (%7.3f ms)  q=%-4s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, short[:60]))
    last_end = max(last_end, e)
