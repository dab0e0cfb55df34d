#!/bin/bash
# usage: ab_multi.sh rounds "ENV1" "ENV2" ... ; interleaved bench runs
R=$1; shift
for i in $(seq $R); do
  for e in "$@"; do
    env $e timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extra --steps 6 --warmup 2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$e', d['ms_per_step'], d['value'])"
  done
done
