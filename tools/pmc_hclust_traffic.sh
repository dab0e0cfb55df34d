#!/bin/bash
# the interpreter itself goes after `--`: a shim script (pyenv, a conda wrapper) would be an exec hop under the profiler's preloaded GPU runtime
PY=$(python3 -c 'import os,sys;print(os.path.realpath(sys.executable))')
# ON THE GPU BOX: HBM bytes of the agglomeration kernel (FETCH_SIZE doubled on gfx950, see tools/pmc_calib.sh) at 375 tasks
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  SHARP_HC_RANGES=1 timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_hct_$c -- "$PY" $REPO/tools/bench_hc.py 15 > $OUT/pmc_hct_$c.log 2>&1
done
cd $REPO
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("gpurun_out/pmc_hct_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "hclust_rnn" in k or "hclust_tri" in k:
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
for k, v in agg.items():
    calls = cnt[(k, "FETCH_SIZE")] / 2.0        # two launches (chunks) per SHARP call
    rd, wr = 2 * v["FETCH_SIZE"] * 1024 / calls, v["WRITE_SIZE"] * 1024 / calls
    n2 = 375 * 2000.0 * 2000.0 * 8
    print("%s: per SHARP call read %.1f GB (%.2f n^2) write %.1f GB (%.2f n^2) total %.1f GB" % (k, rd / 1e9, rd / n2, wr / 1e9, wr / n2, (rd + wr) / 1e9))
PY
