#!/bin/bash
# ON THE GPU BOX: interleaved A/B of one environment switch on ONE box (boxes differ by +-3 ms per cfg2 step).
# usage: tools/ab_env.sh "VAR=a" "VAR=b" [rounds] [bench.py arguments...]
A=$1; B=$2; R=${3:-3}; shift 3
for i in $(seq $R); do
  for e in "$A" "$B"; do
    env $e timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extra "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$e', d['ms_per_step'], d['value'])"
  done
done
