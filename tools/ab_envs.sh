#!/bin/bash
# ON THE GPU BOX: interleaved runs of bench.py under several environment settings on ONE box.
# usage: tools/ab_envs.sh rounds "bench args" "ENV1=a ENV2=b" "ENV1=c" ...      ("-" = no setting)
R=$1; ARGS=$2; shift 2
for i in $(seq $R); do
  for e in "$@"; do
    if [ "$e" = "-" ]; then ee=""; else ee="$e"; fi
    env $ee timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-extra --no-traffic --no-forview $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%-44s' % '$e', d['ms_per_step'], d['value'], d['clusters_found'])"
  done
done
