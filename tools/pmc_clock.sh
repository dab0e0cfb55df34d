#!/bin/bash
# the interpreter itself goes after `--`: a shim script (pyenv, a conda wrapper) would be an exec hop under the profiler's preloaded GPU runtime
PY=$(python3 -c 'import os,sys;print(os.path.realpath(sys.executable))')
# ON THE GPU BOX: the shader clock each kernel actually ran at = GRBM_GUI_ACTIVE (cycles the GPU was busy during the dispatch,
# kernels run one at a time under counter collection; the counter is summed over the 8 XCDs) / 8 / the dispatch's duration from the
# kernel trace of the same run.  Short kernels read high: the busy window includes the dispatch overhead, the duration does not.
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SHARP_HC_PIPE=0 timeout -k 10 400 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_clock -- "$PY" $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra > $OUT/pmc_clock.log 2>&1
cd $REPO
python3 - <<'PY'
import csv, glob, collections
dur = {}
for f in glob.glob("gpurun_out/pmc_clock/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"].split("(")[0], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
for f in glob.glob("gpurun_out/pmc_clock/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or r["Dispatch_Id"] not in dur: continue
        k, d = dur[r["Dispatch_Id"]]
        a = agg[k]; a[0] += float(r["Counter_Value"]); a[1] += d; a[2] += 1
print("%-44s %8s %12s %10s" % ("kernel", "launches", "total ms", "GHz"))
for k, (c, d, n) in sorted(agg.items(), key=lambda x: -x[1][1]):
    if d > 2e5: print("%-44s %8d %12.3f %10.3f" % (k[:44], n, d / 1e6, c / d / 8))
PY
