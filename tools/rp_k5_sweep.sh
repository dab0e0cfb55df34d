#!/bin/bash
# ON THE GPU BOX: grid / chunk sweep of the RP stage at the K = 5 shapes (tools/bench_rp.py configs 1 = a cfg3 block, 5 = cfg4's per-GPU share)
ulimit -c 0
run() { env "$@" timeout -k 10 200 python tools/bench_rp.py $CFG 2>&1 | grep "^m=" | sed 's/proj_build.*rp=/rp=/; s/read+write.*nz=[0-9.]*//' | sed "s/^/$* : /"; }
CFG=1
for cp in 2 3 4 6; do for ap in 2 3 4; do run SHARP_RP_CP_WGS=$cp SHARP_RP_AP_WGS=$ap; done; done
for ch in 4167 6250 12500 16384; do run SHARP_RP_CHUNK=$ch; done
run SHARP_RP_SERIAL=1
CFG=5
for cp in 3 6; do for ap in 2 4; do run SHARP_RP_CP_WGS=$cp SHARP_RP_AP_WGS=$ap; done; done
for ch in 4063 8125 16250; do run SHARP_RP_CHUNK=$ch; done
