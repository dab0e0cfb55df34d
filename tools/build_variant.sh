#!/bin/bash
# tools/build_variant.sh NAME "-DFOO=1 ..." [file.hip | ../../tools/lab/file_lab.hip] [git-rev]: builds sharp_amd/variants/libsharp_hip_NAME.so with one
# translation unit (default rp2.hip) compiled with extra flags, or taken from another git revision, for A/B runs on one box
# (tools/bench_rp.py and tools/bench_hc.py load it when SHARP_VARIANT=NAME).
set -e
cd "$(dirname "$0")/../sharp_amd/csrc"
mkdir -p ../variants
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -pthread"
SRC=${3:-rp2.hip}
IN=$SRC
if [ -n "$4" ]; then git show "$4:sharp_amd/csrc/$SRC" > /tmp/variant_$1_$SRC; cp /tmp/variant_$1_$SRC ./_variant_$1.hip; IN=_variant_$1.hip; fi
/opt/rocm/bin/hipcc $F $2 -c $IN -o /tmp/variant_$1.o
[ -n "$4" ] && rm -f ./_variant_$1.hip
BASE=$(basename $SRC .hip); BASE=${BASE%_lab}          # (a lab copy tools/lab/X_lab.hip stands in for X.hip)
OBJS=$(ls *.o | grep -v "^${BASE}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../variants/libsharp_hip_$1.so $OBJS /tmp/variant_$1.o
echo built $1
