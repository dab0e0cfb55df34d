#!/bin/bash
# the interpreter itself goes after `--`: a shim script (pyenv, a conda wrapper) would be an exec hop under the profiler's preloaded GPU runtime
PY=$(python3 -c 'import os,sys;print(os.path.realpath(sys.executable))')
# ON THE GPU BOX: SQ counters of the RP kernels (serial mode: compact and apply do not overlap), one pass per counter group
# usage: tools/pmc_rp.sh [bench_rp config index, default 0] [output tag]
CFG=${1:-0}; TAG=${2:-rp}
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVES"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  SHARP_RP_SERIAL=1 timeout 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_${TAG}_$tag -- "$PY" $REPO/tools/bench_rp.py $CFG > $OUT/pmc_${TAG}_$tag.log 2>&1
done
cd $REPO
TAG=$TAG python3 - <<'PY'
import csv, glob, collections, os
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in glob.glob("gpurun_out/pmc_%s_*/**/*counter_collection.csv" % os.environ.get("TAG","rp"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "rp_apply" in k or "rp_compact" in k:
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
for k, v in agg.items():
    print(k)
    for c, x in sorted(v.items()):
        print("   %-24s %.4g" % (c, x))
PY
