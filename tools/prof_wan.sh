#!/bin/bash
# rocprofv3 --kernel-trace --stats of a shortened Wan2.1-14B loop:  bash tools/prof_wan.sh <tag> [bench_wan.py args...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o t -- python $R/tools/bench_wan.py "$@" > $OUT/bench.json 2> $OUT/bench.err
cp $OUT/t/t_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
rm -rf $OUT/t
head -5 $OUT/kernel_stats.csv | cut -c1-160
