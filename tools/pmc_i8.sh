#!/bin/bash
# the interpreter itself goes after `--`: a shim script (pyenv, a conda wrapper) would be an exec hop under the profiler's preloaded GPU runtime
PY=$(python3 -c 'import os,sys;print(os.path.realpath(sys.executable))')
# ON THE GPU BOX: counters of dist_i8_kernel (tools/bench_i8.py), one pass per group
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_I8" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_i8_$i -- "$PY" $REPO/tools/bench_i8.py 2000 391 188 2 > $OUT/pmc_i8_$i.log 2>&1
done
cd $REPO
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("gpurun_out/pmc_i8_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        kn = row["Kernel_Name"]
        k = "dist_i8_kernel" if "dist_i8_kernel" in kn else "slice_rows_kernel" if "slice_rows" in kn else "gemm_tn_f64_fast_kernel" if "gemm_tn_f64_fast" in kn else None
        if k:
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k][row["Counter_Name"]] += 1
for k, v in agg.items():
    print(k)
    for c, x in sorted(v.items()):
        print("   %-28s %.5g per launch (%d launches)" % (c, x / max(n[k][c], 1), n[k][c]))
PY
rm -rf gpurun_out/pmc_i8_*/
