#!/bin/bash
# A/B builds of the selection kernel: tools/build_alt_sel.sh NAME "extra hipcc flags" -> alt_libs/NAME.so
cd "$(dirname "$0")/.." || exit 1
mkdir -p alt_libs jenga_amd/build/alt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -c jenga_amd/csrc/select.hip -o jenga_amd/build/alt/select_$1.o \
  -Iinclude -ffp-contract=off -fno-slp-vectorize -Xclang -target-feature -Xclang -packed-fp32-ops -Xclang -target-feature -Xclang -fma-mix-insts $2 2>&1 | grep -v "not a recognized feature" ; [ -f jenga_amd/build/alt/select_$1.o ] || exit 1
OBJS=$(ls jenga_amd/build/*.o | grep -v "/select.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o alt_libs/$1.so $OBJS jenga_amd/build/alt/select_$1.o -lhipblaslt && echo built alt_libs/$1.so
