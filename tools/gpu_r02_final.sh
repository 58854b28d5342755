#!/bin/bash
# round-2 closing measurements (run on the GPU box through gpurun):  bash tools/gpu_r02_final.sh <A|B|T>
#   A: rocprofv3 kernel stats of the default bench, full 50-step loop, the single-GPU presets
#   B: --peaky regime, per-rank work of the 8-GPU presets (--simulate-ranks 8), Wan2.1-14B forward
#   T: the whole -m gpu suite
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r02_final
mkdir -p $O
run() { tag=$1; shift; timeout 900 python bench.py "$@" > $O/$tag.json 2> $O/$tag.err; tail -c 700 $O/$tag.json; echo; tail -2 $O/$tag.err; }
case "$1" in
A)
  bash tools/prof_bench.sh r02_default
  run full50 --steps 50 --warmup 1 --no-cpu-baseline
  run turbo --preset turbo --no-cpu-baseline
  run flash --preset flash --no-cpu-baseline
  run 3stage --preset 3stage --no-cpu-baseline
  run 3stage_i2v --preset 3stage --i2v --no-cpu-baseline
  ;;
A2)
  run default_final
  run full50 --steps 50 --warmup 1 --no-cpu-baseline
  run turbo --preset turbo --no-cpu-baseline
  run flash --preset flash --no-cpu-baseline
  run 3stage --preset 3stage --no-cpu-baseline
  run 3stage_i2v --preset 3stage --i2v --no-cpu-baseline
  run sim8_base --simulate-ranks 8 --steps 3 --no-cpu-baseline
  run sim8_turbo --simulate-ranks 8 --preset turbo-mgpu --steps 3 --no-cpu-baseline
  ;;
B)
  run peaky4 --peaky 4 --no-cpu-baseline
  run peaky8 --peaky 8 --no-cpu-baseline
  run sim8_base --simulate-ranks 8 --steps 3 --no-cpu-baseline
  run sim8_turbo --simulate-ranks 8 --preset turbo-mgpu --steps 3 --no-cpu-baseline
  run sim8_3stage_i2v --simulate-ranks 8 --preset 3stage-mgpu --i2v --steps 3 --no-cpu-baseline
  timeout 600 python tools/bench_wan.py --qk-gain 4 > $O/wan14b_gain4.json 2> $O/wan14b_gain4.err; tail -c 600 $O/wan14b_gain4.json
  ;;
P)
  # board power / clock while each kernel variant loops for ~15 s (is the clock drop a power cap?)
  for v in "lp:9:" "lp_l2:9:--l2-resident 64"; do
    IFS=: read tag fl extra <<< "$v"
    timeout 400 python tools/power_sample.py --out $O/power_$tag.json -- python tools/bench_attn.py --drop 0.7 --iters 600 --attn-only --flags $fl $extra > $O/power_$tag.log 2>&1
    python - $O/power_$tag.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[1], d.get("power_cap_W"), d.get("power_W"), d.get("sclk_MHz"), d.get("stdout_tail","")[-300:])
PY
  done
  ;;
S)
  # sustained (30 s) runs of the three kernels: steady-state power / clock / throughput, flat lists and 85 %-shared lists
  for v in "s_lp:9:" "s_legacy:5:" "s_pair:1:" "s_lp_ov:9:--pair-overlap 0.8" "s_pair_ov:1:--pair-overlap 0.8" "s_legacy_ov:5:--pair-overlap 0.8"; do
    IFS=: read tag fl extra <<< "$v"
    timeout 400 python tools/power_sample.py --out $O/power_$tag.json -- python tools/bench_attn.py --drop 0.7 --iters 500 --attn-only --flags $fl $extra > $O/power_$tag.log 2>&1
    python - $O/power_$tag.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[1], d.get("power_W"), d.get("sclk_MHz"), d.get("stdout_tail","")[-160:])
PY
  done
  ;;
T)
  timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/gpu_tests.log
  ;;
esac
