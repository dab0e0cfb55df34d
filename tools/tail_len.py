"""From the stderr of tools/step_marks.py: per call, ms from the LAST agglomeration's end (device) to the call's return, and the call's length."""
import sys, re, statistics as st
calls, cur = [], []
for line in open(sys.argv[1]):
    if line.startswith("==== call"):
        if cur: calls.append(cur)
        cur = []
    m = re.match(r"\[step\]\s+([0-9.]+) ms\s+(.*)", line)
    if m: cur.append((float(m.group(1)), m.group(2)))
if cur: calls.append(cur)
tails, totals = [], []
for c in calls[2:]:
    ag = [t for t, l in c if "an agglomeration ends" in l]
    ret = [t for t, l in c if "returns" in l]
    if ag and ret: tails.append(ret[-1] - ag[-1]); totals.append(ret[-1])
print("calls %d: after the last agglomeration: min %.2f median %.2f max %.2f ms; call: min %.2f median %.2f ms" % (len(tails), min(tails), st.median(tails), max(tails), min(totals), st.median(totals)))
