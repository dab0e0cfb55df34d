#!/bin/bash
# Run ON THE GPU BOX from the repo root:  bash tools/profile_round6.sh r06
# The default bench (BASELINE.json configs[2]: SHARP_unlimited, 10 blocks) and cfg2 under rocprofv3: kernel stats (one pass), FETCH_SIZE and
# WRITE_SIZE (one pass each: the two TCC counters do not fit one pass; counter collection is never combined with other trace domains),
# then the plain bench line.  tools/profile_collect.py TAG 4 cfg3 turns gpurun_out/TAG_* into the summaries committed under profiles/.
TAG=${1:-r06}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
PY=$(python3 -c 'import os,sys;print(os.path.realpath(sys.executable))')   # the interpreter itself after `--`, never a shim script
cd /tmp && export TMPDIR=/tmp
CMD="$PY $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-traffic --no-forview"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_kt -- $CMD > $OUT/${TAG}_kt.log 2>&1
echo "kt done"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -- $CMD > $OUT/${TAG}_pmc_fetch.log 2>&1
echo "fetch done"
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -- $CMD > $OUT/${TAG}_pmc_write.log 2>&1
echo "write done"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_cfg2_kt -- $CMD --config cfg2 > $OUT/${TAG}_cfg2_kt.log 2>&1
echo "cfg2 kt done"
cd $REPO
# the bench line of the SAME process as the kernel-trace pass: its live (HIP-event) launch time of the roofline kernel is what the
# kernel_stats average has to agree with (events add the launch gap: a few percent)
grep '^{"metric"' $OUT/${TAG}_kt.log | tail -1 > $OUT/${TAG}_bench_under_rocprof.json
grep '^{"metric"' $OUT/${TAG}_cfg2_kt.log | tail -1 > $OUT/${TAG}_cfg2_bench_under_rocprof.json
python3 tools/timeline.py $OUT/${TAG}_cfg2_kt 150 -2 > $OUT/${TAG}_cfg2_step_timeline.txt 2>&1
echo "profiled passes done; now: timeout 1100 python3 bench.py > gpurun_out/${TAG}_bench_n1.json 2> gpurun_out/${TAG}_bench_n1.err  (a call of its own), then python3 tools/profile_collect.py $TAG 4 cfg3"
