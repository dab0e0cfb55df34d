#!/bin/bash
# ON THE GPU BOX: interleaved runs of bench.py with several builds of the library on ONE box (sharp_amd/variants/libsharp_hip_NAME.so; "cur" = the
# committed build), then min / median / mean of ms_per_step per build.   usage: tools/ab_lib_stat.sh rounds "bench args" cur NAME [NAME ...]
R=$1; ARGS=$2; shift 2
cp sharp_amd/libsharp_hip.so /tmp/_cur.so
TMP=$(mktemp)
for i in $(seq $R); do
  for w in "$@"; do
    if [ $w = cur ]; then cp /tmp/_cur.so sharp_amd/libsharp_hip.so; else cp sharp_amd/variants/libsharp_hip_$w.so sharp_amd/libsharp_hip.so; fi
    timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-extra --no-traffic --no-forview $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$w', d['ms_per_step'])" >> $TMP
  done
done
cp /tmp/_cur.so sharp_amd/libsharp_hip.so
python3 - $TMP <<'PY'
import sys, statistics as st
d = {}
for line in open(sys.argv[1]):
    k, v = line.rsplit(None, 1); d.setdefault(k, []).append(float(v))
for k, v in d.items():
    print("%-24s n=%d  min %.2f  median %.2f  mean %.2f  max %.2f" % (k, len(v), min(v), st.median(v), st.mean(v), max(v)))
PY
rm -f $TMP
