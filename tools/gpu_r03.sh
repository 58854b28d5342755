#!/bin/bash
# round-3 GPU sessions (run on the GPU box through gpurun):  bash tools/gpu_r03.sh <A|...>
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r03
mkdir -p $O
run() { tag=$1; shift; timeout 900 python bench.py "$@" > $O/$tag.json 2> $O/$tag.err; tail -c 900 $O/$tag.json; echo; tail -2 $O/$tag.err; }
case "$1" in
A)
  # first session: the new N-rank SP DiT parity test, then the whole suite, the default bench (traffic chain + dense
  # reference), and where the per-rank non-attention time of an 8-rank job goes
  timeout 600 python -m pytest tests/test_gpu_sp_dit.py -x -q -m gpu > $O/A_sp_dit.log 2>&1; tail -15 $O/A_sp_dit.log
  timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_gpu_sp_dit.py > $O/A_suite.log 2>&1; tail -8 $O/A_suite.log
  run A_default
  bash tools/prof_bench.sh r03_sim8 --simulate-ranks 8 --steps 3 --no-cpu-baseline --no-dense-ref > $O/A_prof_sim8.log 2>&1
  head -40 gpurun_out/prof_r03_sim8/kernel_stats.csv
  ;;
esac
