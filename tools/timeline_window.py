"""Kernels longer than min_us in a time window [t0_ms, t1_ms) (relative to the LAST `anchor` kernel launch minus back_ms) of a
rocprofv3 --kernel-trace run: usage timeline_window.py DIR anchor back_ms span_ms [min_us]"""
import csv, glob, os, sys
root, anchor, back, span = sys.argv[1], sys.argv[2], float(sys.argv[3]), float(sys.argv[4])
min_us = float(sys.argv[5]) if len(sys.argv) > 5 else 100.0
files = sorted(glob.glob(root + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = []
for r in csv.DictReader(open(files[-1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]))
rows.sort()
anchors = [r[0] for r in rows if anchor in r[3]]
t0 = anchors[int(__import__("os").environ.get("ANCHOR_IDX", "-1"))] - back * 1e6
for s, e, q, name in rows:
    if s < t0 or s > t0 + span * 1e6 or (e - s) / 1e3 < min_us:
        continue
    short = name.split("(")[0].replace("void ", "").replace("sharp::", "").replace("(anonymous namespace)::", "")
    print("%8.3f -> %8.3f  (%7.3f ms)  q=%-4s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, short[:60]))
