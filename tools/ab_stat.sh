#!/bin/bash
# ON THE GPU BOX: interleaved runs of bench.py under several environment settings on ONE box, then min / median / mean of ms_per_step per setting.
# usage: tools/ab_stat.sh rounds "bench args" "ENV1=a ENV2=b" "ENV1=c" ...      ("-" = no setting)
R=$1; ARGS=$2; shift 2
TMP=$(mktemp)
for i in $(seq $R); do
  for e in "$@"; do
    if [ "$e" = "-" ]; then ee=""; else ee="$e"; fi
    env $ee timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-extra --no-traffic --no-forview $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$e', d['ms_per_step'])" >> $TMP
  done
done
python3 - $TMP <<'PY'
import sys, statistics as st
d = {}
for line in open(sys.argv[1]):
    k, v = line.rsplit(None, 1); d.setdefault(k, []).append(float(v))
for k, v in d.items():
    print("%-48s n=%d  min %.2f  median %.2f  mean %.2f  max %.2f" % (k, len(v), min(v), st.median(v), st.mean(v), max(v)))
PY
rm -f $TMP
