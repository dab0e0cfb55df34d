#!/bin/bash
# tools/build_variant.sh NAME "-DFOO=1 ..." : builds sharp_amd/variants/libsharp_hip_NAME.so with extra flags on rp2.hip
set -e
cd "$(dirname "$0")/../sharp_amd/csrc"
mkdir -p ../variants
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -pthread"
/opt/rocm/bin/hipcc $F $2 -c rp2.hip -o /tmp/rp2_$1.o
OBJS=$(ls *.o | grep -v '^rp2.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../variants/libsharp_hip_$1.so $OBJS /tmp/rp2_$1.o
echo built $1
