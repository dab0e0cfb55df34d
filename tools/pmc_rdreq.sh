#!/bin/bash
# the interpreter itself goes after `--`: a shim script (pyenv, a conda wrapper) would be an exec hop under the profiler's preloaded GPU runtime
PY=$(python3 -c 'import os,sys;print(os.path.realpath(sys.executable))')
# ON THE GPU BOX: exact HBM-side read bytes from the L2's request-size counters (32 / 64 / 128 B), calibrated on known-size streams
# and applied to the agglomeration kernel at 375 tasks.
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
C="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
timeout 300 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_rq_calib -- $REPO/tools/micro/fetch_calib 1024 > $OUT/pmc_rq_calib.log 2>&1
SHARP_HC_RANGES=1 timeout 300 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_rq_hc -- "$PY" $REPO/tools/bench_hc.py 15 > $OUT/pmc_rq_hc.log 2>&1
cd $REPO
python3 - <<'PY'
import csv, glob, collections
for tag in ("calib", "hc"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("gpurun_out/pmc_rq_%s/**/*counter_collection.csv" % tag, recursive=True):
        for row in csv.DictReader(open(f)):
            agg[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in agg.items():
        if tag == "hc" and "hclust_rnn_kernel<1024, 0>" not in k: continue
        if tag == "calib" and "stream" not in k: continue
        n = len(v["TCC_EA0_RDREQ_sum"])
        tot = sum(v["TCC_EA0_RDREQ_sum"]); r32 = sum(v["TCC_EA0_RDREQ_32B_sum"]); r64 = sum(v["TCC_EA0_RDREQ_64B_sum"]); r128 = sum(v["TCC_EA0_RDREQ_128B_sum"])
        by = r32 * 32 + r64 * 64 + r128 * 128
        print("%-52s launches %3d  RDREQ %.4g  32B %.4g  64B %.4g  128B %.4g  => %.3f GB per launch (other-size requests: %.4g)" % (k[:52], n, tot, r32, r64, r128, by / n / 1e9, tot - r32 - r64 - r128))
PY
tail -3 $OUT/pmc_rq_hc.log
