#!/bin/bash
# ON THE GPU BOX: interleaved A/B of two builds of the library on ONE box: sharp_amd/variants/libsharp_hip_$1.so against the current one.
# usage: tools/ab_lib.sh VARIANT [rounds] [bench.py arguments...]
V=$1; R=${2:-3}; shift 2
cp sharp_amd/libsharp_hip.so /tmp/_cur.so
for i in $(seq $R); do
  for w in $V cur; do
    if [ $w = cur ]; then cp /tmp/_cur.so sharp_amd/libsharp_hip.so; else cp sharp_amd/variants/libsharp_hip_$V.so sharp_amd/libsharp_hip.so; fi
    timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-extra "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$w', d['ms_per_step'], d['value'])"
  done
done
cp /tmp/_cur.so sharp_amd/libsharp_hip.so
