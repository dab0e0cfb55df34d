#!/bin/bash
# ON THE GPU BOX: FETCH_SIZE / WRITE_SIZE of known-size streaming kernels (calibration of the gfx950 counter semantics)
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
timeout 120 $REPO/tools/micro/fetch_calib 4096 > $OUT/fetch_calib.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_calib_$c -- $REPO/tools/micro/fetch_calib 1024 > $OUT/pmc_calib_$c.log 2>&1
done
cd $REPO
cat $OUT/fetch_calib.txt
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_calib_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in agg.items():
    for c, xs in sorted(v.items()):
        print("%-60s %-11s first two launches (1 GiB each): %s" % (k[:60], c, ["%.4g" % x for x in xs[:2]]))
PY
