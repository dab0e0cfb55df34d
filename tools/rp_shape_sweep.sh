#!/bin/bash
# ON THE GPU BOX: workgroup-shape variants of the producer / consumer RP kernel (built in the container by tools/build_variant.sh ... rp3.hip into
# sharp_amd/variants/) at the K = 5 shapes: bench_rp.py config 1 = a cfg3 block, 5 = cfg4's per-GPU share; interleaved rounds on one box.
# usage: tools/rp_shape_sweep.sh "variant names" [rounds]      ("-" = the committed library)
ulimit -c 0
R=${2:-2}
for r in $(seq 1 $R); do
  for v in $1; do
    for cfg in 1 5; do
      if [ "$v" = "-" ]; then env -u SHARP_VARIANT timeout -k 10 200 python tools/bench_rp.py $cfg 2>&1 | grep "^m=" | sed 's/proj_build.*rp=/rp=/; s/read+write.*nz=[0-9.]*//' | sed "s/^/committed : /"
      else SHARP_VARIANT=$v timeout -k 10 200 python tools/bench_rp.py $cfg 2>&1 | grep "^m=" | sed 's/proj_build.*rp=/rp=/; s/read+write.*nz=[0-9.]*//' | sed "s/^/$v : /"; fi
    done
  done
done
