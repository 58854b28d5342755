#!/bin/bash
# A/B builds of the pair kernel: tools/build_alt5.sh NAME "extra hipcc flags" -> alt_libs/NAME.so (the other objects are the
# tree's: run `python -m jenga_amd.build` first)
cd "$(dirname "$0")/.." || exit 1
mkdir -p alt_libs jenga_amd/build/alt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -c jenga_amd/csrc/bsattn5.hip -o jenga_amd/build/alt/bsattn5_$1.o \
  -fno-honor-nans -fno-slp-vectorize -Wno-inline-asm $2 || exit 1
OBJS=$(ls jenga_amd/build/*.o | grep -v bsattn5.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o alt_libs/$1.so $OBJS jenga_amd/build/alt/bsattn5_$1.o -lhipblaslt && echo built alt_libs/$1.so
