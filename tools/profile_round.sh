#!/bin/bash
# Run ON THE GPU BOX from the repo root:  bash tools/profile_round.sh r01
# Produces gpurun_out/<tag>_* ; tools/profile_collect.py turns them into the summaries committed under profiles/.
# Three separate rocprofv3 passes (kernel stats; FETCH_SIZE; WRITE_SIZE): the two TCC counters do not fit one pass, and
# counter collection is never combined with other trace domains.
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PY=$(python3 -c 'import os,sys;print(os.path.realpath(sys.executable))')   # the interpreter itself after `--`, never a shim script
CMD="$PY $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-traffic"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_kt -- $CMD > $OUT/${TAG}_kt.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -- $CMD > $OUT/${TAG}_pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -- $CMD > $OUT/${TAG}_pmc_write.log 2>&1
cd $REPO
# the bench line of the SAME process as the kernel-trace pass: its live (HIP-event) launch time of the roofline kernel is what the
# kernel_stats average has to agree with (events add the launch gap: a few percent)
grep '^{"metric"' $OUT/${TAG}_kt.log | tail -1 > $OUT/${TAG}_bench_under_rocprof.json
timeout 600 python3 bench.py --steps 5 --warmup 2 > $OUT/${TAG}_bench_n1.json 2> $OUT/${TAG}_bench_n1.err
tail -c 600 $OUT/${TAG}_bench_n1.json
