#!/bin/bash
# A/B build of ONE source of the library: tools/build_alt_one.sh NAME SOURCE.hip "extra hipcc flags" -> alt_libs/NAME.so
cd "$(dirname "$0")/.." || exit 1
mkdir -p alt_libs jenga_amd/build/alt
B=$(basename "$2" .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -c jenga_amd/csrc/$B.hip -o jenga_amd/build/alt/${B}_$1.o \
  -ffp-contract=off -Iinclude $3 || exit 1
OBJS=$(ls jenga_amd/build/*.o | grep -v "/$B.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o alt_libs/$1.so $OBJS jenga_amd/build/alt/${B}_$1.o -lhipblaslt && echo built alt_libs/$1.so
