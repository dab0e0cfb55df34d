"""Kernel timeline of the LAST step of a bench.py run under rocprofv3 --kernel-trace: usage timeline_step.py DIR [min_us] [t_from_ms] [t_to_ms]
(ms from the step's projector draw kernel; kernels shorter than min_us are left out)."""
import csv, glob, os, sys
root = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
t_from = float(sys.argv[3]) if len(sys.argv) > 3 else -1e9
t_to = float(sys.argv[4]) if len(sys.argv) > 4 else 1e9
files = sorted(glob.glob(root + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = []
for r in csv.DictReader(open(files[-1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]))
rows.sort()
draws = [r[0] for r in rows if "proj_draw_kernel" in r[3]]
t0 = draws[-1]
print("# last step: %.2f ms from the projector draw to the end of the last kernel" % ((max(r[1] for r in rows) - t0) / 1e6))
for s, e, q, name in rows:
    t = (s - t0) / 1e6
    if s < t0 or t < t_from or t > t_to or (e - s) / 1e3 < min_us:
        continue
    short = name.split("(")[0].replace("void ", "").replace("sharp::", "").replace("(anonymous namespace)::", "")
    print("%8.3f -> %8.3f  (%7.3f ms)  q=%-3s %s" % (t, (e - t0) / 1e6, (e - s) / 1e6, q, short[:60]))
