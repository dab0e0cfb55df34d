#!/bin/bash
# the interpreter itself goes after `--`: a shim script (pyenv, a conda wrapper) would be an exec hop under the profiler's preloaded GPU runtime
PY=$(python3 -c 'import os,sys;print(os.path.realpath(sys.executable))')
# ON THE GPU BOX: LDS bank-conflict counters of rp_apply for the current library and for a variant (tools/build_variant.sh NAME)
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in "" "$1"; do
  tag=${v:-cur}
  SHARP_VARIANT=$v SHARP_RP_SERIAL=1 timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --output-format csv -d $OUT/pmc_banks_$tag -- "$PY" $REPO/tools/bench_rp.py ${2:-0} > $OUT/pmc_banks_$tag.log 2>&1
done
cd $REPO
python3 - <<'PY'
import csv, glob, collections
for tag in sorted(glob.glob("gpurun_out/pmc_banks_*/")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(tag + "**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            if "rp_apply" in k:
                agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    for k, v in agg.items():
        print(tag, k, {c: "%.4g" % x for c, x in sorted(v.items())})
PY
