#!/bin/bash
# the interpreter itself goes after `--`: a shim script (pyenv, a conda wrapper) would be an exec hop under the profiler's preloaded GPU runtime
PY=$(python3 -c 'import os,sys;print(os.path.realpath(sys.executable))')
# ON THE GPU BOX: SQ counters of the producer / consumer RP kernel (rp3.hip), one pass per counter group
# usage: tools/pmc_pc.sh [bench_rp config index, default 1] [output tag]
CFG=${1:-1}; TAG=${2:-pc}
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_${TAG}_$tag -- "$PY" $REPO/tools/bench_rp.py $CFG > $OUT/pmc_${TAG}_$tag.log 2>&1
done
cd $REPO
TAG=$TAG python3 - <<'PY'
import csv, glob, collections, os
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("gpurun_out/pmc_%s_*/**/*counter_collection.csv" % os.environ.get("TAG","pc"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "rp_pc_kernel" in k:
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k][row["Counter_Name"]] += 1
for k, v in agg.items():
    print(k)
    for c, x in sorted(v.items()):
        print("   %-24s %.5g per launch (%d launches)" % (c, x / max(n[k][c], 1), n[k][c]))
PY
