#!/bin/bash
# the interpreter itself goes after `--`: a shim script (pyenv, a conda wrapper) would be an exec hop under the profiler's preloaded GPU runtime
PY=$(python3 -c 'import os,sys;print(os.path.realpath(sys.executable))')
# ON THE GPU BOX, from the repo root:  bash tools/profile_rp_k5.sh r03
# The RP matmul stage alone at the K = 5 shapes (bench_rp.py configs 1 = a cfg3 block, 5 = cfg4's per-GPU share): a kernel trace with
# statistics, then FETCH_SIZE and WRITE_SIZE in passes of their own (never combined with other trace domains).  tools/profile_collect_rp.py
# turns the output into profiles/<tag>_rp_k5_*.
TAG=${1:-r03}
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in 1 5; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_rpk5_kt_$cfg -- "$PY" $REPO/tools/bench_rp.py $cfg > $OUT/${TAG}_rpk5_kt_$cfg.log 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_rpk5_fetch_$cfg -- "$PY" $REPO/tools/bench_rp.py $cfg > $OUT/${TAG}_rpk5_fetch_$cfg.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_rpk5_write_$cfg -- "$PY" $REPO/tools/bench_rp.py $cfg > $OUT/${TAG}_rpk5_write_$cfg.log 2>&1
done
cd $REPO
python3 tools/profile_collect_rp.py $TAG
