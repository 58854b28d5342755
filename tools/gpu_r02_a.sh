#!/bin/bash
# round 2, GPU call A: correctness of the pair kernel first, then speed (round-1 kernel vs pair kernel), then the suite
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r02a
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_pair.py -x -q 2>&1 | tail -25 > $O/pair_tests.log
cat $O/pair_tests.log
for fl in 5 1; do
  timeout 300 python tools/bench_attn.py --iters 3 --drop 0.7 --flags $fl > $O/attn_flat_f$fl.json 2> $O/attn_flat_f$fl.err
  timeout 300 python tools/bench_attn.py --iters 3 --drop 0.7 --flags $fl --pair-overlap 0.8 > $O/attn_ov80_f$fl.json 2> $O/attn_ov80_f$fl.err
done
cat $O/attn_*.json
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/gpu_tests.log
cat $O/gpu_tests.log
