#!/bin/bash
# ON THE GPU BOX: interleaved A/B of bench.py under different environment settings, REPS rounds.
# usage: tools/ab_bench.sh REPS "NAME1:VAR=val VAR2=val" "NAME2:" ...   -> gpurun_out/ab_<NAME>.txt (ms per step of each round)
REPS=$1; shift
mkdir -p gpurun_out
for spec in "$@"; do rm -f gpurun_out/ab_${spec%%:*}.txt; done
for r in $(seq $REPS); do
  for spec in "$@"; do
    name=${spec%%:*}; envs=${spec#*:}
    env $envs python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['clusters_found'])" >> gpurun_out/ab_$name.txt
  done
done
for spec in "$@"; do name=${spec%%:*}; echo "$name: $(tr '\n' ' ' < gpurun_out/ab_$name.txt)"; done
