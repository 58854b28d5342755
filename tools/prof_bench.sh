#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench command (run on the GPU box through gpurun):
#   bash tools/prof_bench.sh <tag> [bench.py args...]   -> gpurun_out/prof_<tag>/{kernel_stats.csv, bench.json}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o t -- python $R/bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err
cp $OUT/t/t_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
rm -rf $OUT/t
head -25 $OUT/kernel_stats.csv
tail -c 400 $OUT/bench.json
