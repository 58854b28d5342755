#!/bin/bash
# Like build_alt.sh but for any single source: tools/build_alt_src.sh NAME SRCFILE(select|rowops|bsattn|gilbert) [flags...]
set -e
cd "$(dirname "$0")/.."
NAME=$1; WHICH=$2; shift; shift
mkdir -p alt_libs
EXTRA=""
case $WHICH in select|rowops) EXTRA="-ffp-contract=off";; bsattn) EXTRA="-fno-honor-nans";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -Ijenga_amd/csrc -c jenga_amd/csrc/$WHICH.hip -o alt_libs/$NAME.o $EXTRA "$@"
OBJS=""
for o in capi gilbert rowops select bsattn; do if [ $o = $WHICH ]; then OBJS="$OBJS alt_libs/$NAME.o"; else OBJS="$OBJS jenga_amd/build/$o.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o alt_libs/$NAME.so $OBJS
rm alt_libs/$NAME.o
echo alt_libs/$NAME.so
