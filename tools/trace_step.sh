#!/bin/bash
# Run ON THE GPU BOX from the repo root:  bash tools/trace_step.sh TAG [ENV=VALUE ...]
# One rocprofv3 kernel trace of the default bench (3 timed steps) with the given switches; tools/timeline.py gpurun_out/TAG prints a step.
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
PY=$(python3 -c 'import os,sys;print(os.path.realpath(sys.executable))')
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$TAG -- $PY $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-traffic --no-forview > $OUT/$TAG.log 2>&1 || exit 1
cd $REPO
python3 tools/timeline.py $OUT/$TAG 300 -2 > $OUT/${TAG}_timeline.txt
grep '^{"metric"' $OUT/$TAG.log | tail -1 | cut -c1-200
