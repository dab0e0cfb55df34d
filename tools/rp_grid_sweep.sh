ulimit -c 0
for cfg in "8 4 x" "4 4 0" "2 4 0" "2 2 0" "1 2 0" "3 2 0" "2 3 0" "4 2 0" "2 4 1" "4 4 1"; do
  set -- $cfg
  if [ "$3" = "x" ]; then unset SHARP_RP_SERIAL; else export SHARP_RP_SERIAL=$3; fi
  echo "== CP_WGS=$1 AP_WGS=$2 SERIAL=$3"
  SHARP_RP_CP_WGS=$1 SHARP_RP_AP_WGS=$2 timeout -k 10 200 python tools/bench_rp.py 0 1 2 2>&1 | grep "^m=" | sed 's/proj_build.*rp=/rp=/; s/read+write.*nz=[0-9.]*//'
done
