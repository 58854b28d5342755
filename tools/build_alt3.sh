#!/bin/bash
# Alternative libjenga_amd.so with a variant of the LP kernel (bsattn3.hip): tools/build_alt3.sh NAME [flags...]
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p alt_libs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -Ijenga_amd/csrc -c jenga_amd/csrc/bsattn3.hip -o alt_libs/$NAME.o -fno-honor-nans -fno-slp-vectorize -Wno-inline-asm "$@" 2>&1 | grep -E "error" || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o alt_libs/$NAME.so alt_libs/$NAME.o jenga_amd/build/capi.o jenga_amd/build/gilbert.o jenga_amd/build/rowops.o jenga_amd/build/select.o jenga_amd/build/bsattn.o jenga_amd/build/gemm.o -lhipblaslt
rm alt_libs/$NAME.o
echo alt_libs/$NAME.so
