#!/bin/bash
# Build an alternative libjenga_amd.so for same-box A/B runs (select it with JENGA_LIB=...):
#   tools/build_alt.sh NAME [bsattn source (default: the tree's)] [extra hipcc flags for bsattn...]
# Output: alt_libs/NAME.so (git-ignored, travels with gpurun).  Needs jenga_amd/build/*.o from a normal build.
set -e
cd "$(dirname "$0")/.."
NAME=$1; SRC=${2:-jenga_amd/csrc/bsattn.hip}; shift; shift || true
mkdir -p alt_libs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -Ijenga_amd/csrc -c "$SRC" -o alt_libs/$NAME.o -fno-honor-nans "$@"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o alt_libs/$NAME.so alt_libs/$NAME.o jenga_amd/build/capi.o jenga_amd/build/gilbert.o jenga_amd/build/rowops.o jenga_amd/build/select.o
rm alt_libs/$NAME.o
echo alt_libs/$NAME.so
