#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r02k
mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/gpu_tests.log
cat $O/gpu_tests.log
timeout 600 python bench.py --steps 6 --warmup 1 --no-cpu-baseline > $O/bench_lp.json 2> $O/bench_lp.err; tail -c 1500 $O/bench_lp.json
JENGA_ATTN_FLAGS=5 timeout 600 python bench.py --steps 6 --warmup 1 --no-cpu-baseline > $O/bench_legacy.json 2> $O/bench_legacy.err; tail -c 600 $O/bench_legacy.json
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --simulate-ranks 8 > $O/bench_sim8.json 2> $O/bench_sim8.err; tail -c 600 $O/bench_sim8.json; tail -5 $O/bench_sim8.err
