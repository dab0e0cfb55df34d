#!/bin/bash
# ON THE GPU BOX, from the repo root: bash tools/pmc_round6.sh [rp|hc|all]
#  rp: SQ / LDS counters of rp_pc_kernel on one cfg3 block -- fp32 counts (K = 5 table path) and fp64 CPM doubles (the general path)
#  hc: one cfg3 step (SHARP_unlimited, ten blocks) with the default agglomeration and with the upper-triangle kernel (lab build, SHARP_HC_TRI=1):
#      per kernel FETCH_SIZE, WRITE_SIZE, SQ_WAIT_INST_ANY, SQ_BUSY_CYCLES, SQ_WAVE_CYCLES -- is the chunk period bytes or per-CU latency?
# One rocprofv3 pass per counter group (counter collection is never combined with other trace domains); summaries on stdout.
WHAT=${1:-all}
PY=$(python3 -c 'import os,sys;print(os.path.realpath(sys.executable))')   # the interpreter itself after `--`, never a shim script
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ $WHAT = rp ] || [ $WHAT = all ]; then
  for kind in f32 f64; do
    for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM"; do
      tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
      timeout 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc6_rp_${kind}_$tag -- "$PY" $REPO/tools/rp_one.py $kind 6 > $OUT/pmc6_rp_${kind}_$tag.log 2>&1
      echo "rp $kind $tag done: $(tail -1 $OUT/pmc6_rp_${kind}_$tag.log)"
    done
  done
fi
if [ $WHAT = hc ] || [ $WHAT = all ]; then
  for tri in 0 1; do
    for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_WAVES"; do
      tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
      SHARP_VARIANT=lab SHARP_HC_TRI=$tri timeout 400 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc6_hc_tri${tri}_$tag -- "$PY" $REPO/tools/cfg3_profile.py > $OUT/pmc6_hc_tri${tri}_$tag.log 2>&1
      echo "hc tri=$tri $tag done: $(grep '^call' $OUT/pmc6_hc_tri${tri}_$tag.log | tail -1)"
    done
  done
fi
cd $REPO
python3 - <<'PY'
import csv, glob, collections, os, re
def collect(pattern, want):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.Counter())
    for f in glob.glob(pattern, recursive=True):
        for row in csv.DictReader(open(f)):
            k = re.sub(r"\(.*$", "", row["Kernel_Name"])
            k = re.sub(r"^void ", "", k)
            if any(w in k for w in want):
                agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
    return agg, cnt
for kind in ("f32", "f64"):
    agg, cnt = collect("gpurun_out/pmc6_rp_%s_*/**/*counter_collection.csv" % kind, ("rp_pc_kernel",))
    for k, v in agg.items():
        n = max(cnt[k].values())
        print("== rp_pc_kernel on one cfg3 block, %s values: %s (%d launches; per launch)" % (kind, k[:90], n))
        for c, x in sorted(v.items()):
            print("   %-24s %.4g" % (c, x / cnt[k][c]))
        wc = v.get("SQ_WAVE_CYCLES", 0) / max(cnt[k].get("SQ_WAVE_CYCLES", 1), 1)
        bc = v.get("SQ_BUSY_CYCLES", 0) / max(cnt[k].get("SQ_BUSY_CYCLES", 1), 1)
        def per(c): return v.get(c, 0) / max(cnt[k].get(c, 1), 1)
        if wc:
            print("   -> waves waiting (SQ_WAIT_ANY / SQ_WAVE_CYCLES) %.2f, waiting on an instruction %.2f, issuing %.2f" % (per("SQ_WAIT_ANY") / wc, per("SQ_WAIT_INST_ANY") / wc, per("SQ_ACTIVE_INST_ANY") / wc))
        if bc:
            print("   -> LDS index active / busy cycles %.2f, bank conflicts / LDS active %.2f, VALU active / busy %.2f (4 SIMDs: / 4 = %.2f)" % (
                per("SQ_LDS_IDX_ACTIVE") / bc, per("SQ_LDS_BANK_CONFLICT") / max(per("SQ_LDS_IDX_ACTIVE"), 1), per("SQ_ACTIVE_INST_VALU") / bc, per("SQ_ACTIVE_INST_VALU") / bc / 4))
for tri in (0, 1):
    agg, cnt = collect("gpurun_out/pmc6_hc_tri%d_*/**/*counter_collection.csv" % tri, ("hclust_rnn", "hclust_tri", "gemm_tn_f64_fast", "stats_lane", "rp_pc_kernel"))
    if not agg: continue
    print("== one cfg3 step (two calls profiled: per CALL), SHARP_HC_TRI=%d" % tri)
    for k, v in sorted(agg.items()):
        calls = 2.0
        rd, wr = 2 * v.get("FETCH_SIZE", 0) * 1024 / calls, v.get("WRITE_SIZE", 0) * 1024 / calls
        wc, bc = v.get("SQ_WAVE_CYCLES", 0) / calls, v.get("SQ_BUSY_CYCLES", 0) / calls
        print("   %-60s launches %3d  read %7.1f GB write %6.1f GB | wave cycles %.3g: waiting on an instruction %.2f, any wait %.2f, issuing %.2f (VMEM %.2f VALU %.2f) | busy cycles %.3g" % (
            k[:60], int(max(cnt[k].values()) / calls), rd / 1e9, wr / 1e9, wc, v.get("SQ_WAIT_INST_ANY", 0) / calls / max(wc, 1), v.get("SQ_WAIT_ANY", 0) / calls / max(wc, 1),
            v.get("SQ_ACTIVE_INST_ANY", 0) / calls / max(wc, 1), v.get("SQ_ACTIVE_INST_VMEM", 0) / calls / max(wc, 1), v.get("SQ_ACTIVE_INST_VALU", 0) / calls / max(wc, 1), bc))
PY
