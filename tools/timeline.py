"""Kernel timeline of one SHARP() step from a `rocprofv3 --kernel-trace --output-format csv` run of bench.py:
start / end (ms, relative to the step's first kernel), queue and short name of every kernel longer than --min-us.
usage: timeline.py TRACE_DIR [min_us] [step index among the steps found; default -2 = the last timed step]"""
import csv
import glob
import sys

root = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 150.0
pick = int(sys.argv[3]) if len(sys.argv) > 3 else -2
import os

rows = []
files = sorted(glob.glob(root + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
for f in files[-1:]:                                      # the newest run only (gpurun_out/ accumulates earlier ones)
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if "proj_draw_kernel" in r[3]] + [len(rows)]   # a step begins with the projector draw
pick = pick % (len(starts) - 1)
i0, i1 = starts[pick], starts[pick + 1]
t0 = rows[i0][0]
for s, e, q, name in rows[i0:i1]:
    if (e - s) / 1e3 < min_us:
        continue
    short = name.split("(")[0].replace("void ", "").replace("sharp::", "").replace("(anonymous namespace)::", "")
    print("%8.3f -> %8.3f  (%7.3f ms)  q=%-4s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, short))
