#!/bin/bash
# round-4 GPU sessions (run on the GPU box through gpurun):  bash tools/gpu_r04.sh <A|...>
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r04
mkdir -p $O
run() { tag=$1; shift; timeout 900 python bench.py "$@" > $O/$tag.json 2> $O/$tag.err; tail -c 600 $O/$tag.json; echo; tail -2 $O/$tag.err; }
brief() { python - "$@" <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d["value"], d["config"]["ms_per_class"], d["roofline"]["frac"], d["config"].get("sim_exchange"))
    except Exception as e:
        print(f, "FAILED", e)
PY
}
case "$1" in
A)
  # exchange / compute overlap of the sequence-parallel blocks: parity (thread-simulated ranks, RCCL world 1 harness),
  # then rank 0 of an 8-rank job on one GPU with the exchanges as side-stream delays at a stated xGMI rate
  timeout 900 python -m pytest tests/test_gpu_sp_dit.py tests/test_gpu_rccl.py tests/test_gpu_fused.py tests/test_gpu_ulysses.py -x -q -m gpu > $O/A_sp.log 2>&1; tail -15 $O/A_sp.log
  S="--simulate-ranks 8 --steps 6 --no-cpu-baseline --no-dense-ref"
  JENGA_SP_OVERLAP=0 run A_s8_ov0_x0 $S
  JENGA_SP_OVERLAP=0 run A_s8_ov0_x300 $S --sim-exchange-gbps 300
  run A_s8_ov1_x300 $S --sim-exchange-gbps 300
  run A_s8_ov1_x0 $S
  run A_s8_ov1_x200 $S --sim-exchange-gbps 200
  JENGA_SP_MLP_TAIL=0 run A_s8_ov1_tail0_x300 $S --sim-exchange-gbps 300
  JENGA_SP_MLP_TAIL=0.5 run A_s8_ov1_tail50_x300 $S --sim-exchange-gbps 300
  run A_default --no-cpu-baseline --no-dense-ref
  brief $O/A_*.json
  timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_gpu_sp_dit.py --deselect tests/test_gpu_rccl.py --deselect tests/test_gpu_fused.py --deselect tests/test_gpu_ulysses.py > $O/A_suite.log 2>&1; tail -8 $O/A_suite.log
  ;;
B)
  # the whole suite at ABI 3 (cross-attention entry, choice export / import, RCCL harness), the experiment kernels once
  # through libjenga_amd_exp.so, the default line with roofline_secondary / loop / power / extra.wan14b, the Wan line,
  # the overlap runs with the Q|K / V split in the single-stream blocks
  timeout 1500 python -m pytest tests -q -m gpu -x > $O/B_suite.log 2>&1; tail -12 $O/B_suite.log
  JENGA_LIB=$PWD/jenga_amd/libjenga_amd_exp.so timeout 900 python -m pytest tests/test_gpu_pair.py tests/test_gpu_parity.py -q -m gpu -k "pair or sparse_kernel_vs_oracle" > $O/B_exp.log 2>&1; tail -5 $O/B_exp.log
  S="--simulate-ranks 8 --steps 6 --no-cpu-baseline --no-dense-ref"
  JENGA_SP_OVERLAP=0 run B_s8_ov0_x300 $S --sim-exchange-gbps 300
  run B_s8_ov1_x300 $S --sim-exchange-gbps 300
  run B_s8_ov1_x400 $S --sim-exchange-gbps 400
  run B_s8_ov1_x200 $S --sim-exchange-gbps 200
  JENGA_SP_MLP_TAIL=0.375 run B_s8_ov1_tail37_x300 $S --sim-exchange-gbps 300
  run B_s8_ov1_x0 $S
  run B_default
  brief $O/B_s8*.json $O/B_default.json
  timeout 900 python bench.py --workload wan14b > $O/B_wan14b.json 2> $O/B_wan14b.err; tail -c 1500 $O/B_wan14b.json; tail -3 $O/B_wan14b.err
  ;;
C)
  # cross-attention with a ragged context, choice export / import, the cohort experiment (parity, same-box A/B against the
  # default order on flat and coherent lists, counters), the LDS-flag handoff micro-benchmark
  timeout 900 python -m pytest tests/test_gpu_wan_dit.py tests/test_gpu_order.py tests/test_gpu_fused.py tests/test_gpu_parity.py -x -q -m gpu > $O/C_tests.log 2>&1; tail -8 $O/C_tests.log
  for a in "4000 32 0" "4000 32 8" "4000 16 0" "4000 16 6"; do ./tools/micro/lds_flag_handoff $a; done > $O/C_lds_flag_handoff.txt 2>&1; cat $O/C_lds_flag_handoff.txt
  ba() { tag=$1; shift; timeout 600 python tools/bench_attn.py "$@" > $O/C_attn_$tag.json 2> $O/C_attn_$tag.err; python - $O/C_attn_$tag.json $tag <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], {k: (round(d[k],3) if isinstance(d[k],float) else d[k]) for k in ("attn_ms","attn_TFLOPs","kept_mean","adjacent_shared_frac","flags","finite") if k in d})
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  }
  A="--drop 0.7 --iters 100 --attn-only"
  ba flat_base $A --flags 25
  ba flat_cohort $A --flags 57
  JENGA_COHORT_SIZE=32 ba flat_cohort32 $A --flags 57
  JENGA_COHORT_TIMEOUT_US=1000 ba flat_cohort_t1000 $A --flags 57
  ba flat_cohort_plainorder $A --flags 41
  ba flat_base2 $A --flags 25
  ba coh3_base $A --coherent 3 --gain 2 --flags 25
  ba coh3_cohort $A --coherent 3 --gain 2 --flags 57
  ba coh3_cohort_plainorder $A --coherent 3 --gain 2 --flags 41
  bash tools/pmc_attn2.sh r04_cohort --drop 0.7 --iters 2 --attn-only --flags 57 > $O/C_pmc_cohort.log 2>&1; tail -40 $O/C_pmc_cohort.log
  ;;
D)
  # cohort follow-up: a quorum instead of the full generation (the stragglers start late), shorter timeouts
  ba() { tag=$1; shift; timeout 600 python tools/bench_attn.py "$@" > $O/D_attn_$tag.json 2> $O/D_attn_$tag.err; python - $O/D_attn_$tag.json $tag <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], {k: (round(d[k],3) if isinstance(d[k],float) else d[k]) for k in ("attn_ms","attn_TFLOPs","kept_mean","adjacent_shared_frac","flags","finite") if k in d})
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  }
  A="--drop 0.7 --iters 100 --attn-only"
  ba flat_base $A --flags 25
  JENGA_COHORT_QUORUM=48 ba flat_q48 $A --flags 57
  JENGA_COHORT_QUORUM=32 ba flat_q32 $A --flags 57
  JENGA_COHORT_QUORUM=16 ba flat_q16 $A --flags 57
  JENGA_COHORT_QUORUM=48 JENGA_COHORT_TIMEOUT_US=100 ba flat_q48_t100 $A --flags 57
  JENGA_COHORT_QUORUM=32 JENGA_COHORT_TIMEOUT_US=60 ba flat_q32_t60 $A --flags 57
  JENGA_COHORT_SIZE=128 JENGA_COHORT_QUORUM=64 ba flat_s128_q64 $A --flags 57
  ba flat_base2 $A --flags 25
  JENGA_COHORT_QUORUM=32 ba coh3_q32 $A --coherent 3 --gain 2 --flags 57
  JENGA_COHORT_QUORUM=32 bash tools/pmc_attn2.sh r04_cohort_q32 --drop 0.7 --iters 2 --attn-only --flags 57 > $O/D_pmc_cohort_q32.log 2>&1; grep -A12 '"derived"' $O/D_pmc_cohort_q32.log | head -30
  JENGA_SELECT_FLAGS=1 run D_default_device_scan --no-cpu-baseline --no-dense-ref --no-secondary --no-wan-extra
  run D_default_again --no-cpu-baseline --no-dense-ref --no-secondary --no-wan-extra
  brief $O/D_default*.json
  ;;
E)
  # validation of the round's state: the whole suite, the default line, the overlap efficiency at three exchange rates with
  # the shipped defaults, the multi-rank launch path with a world of one rank, kernel stats of the default command
  timeout 1500 python -m pytest tests -q -m gpu > $O/E_suite.log 2>&1; tail -6 $O/E_suite.log
  S="--simulate-ranks 8 --steps 6 --no-cpu-baseline --no-dense-ref"
  run E_default
  run E_s8_x0 $S
  run E_s8_x300 $S --sim-exchange-gbps 300
  run E_s8_x400 $S --sim-exchange-gbps 400
  run E_s8_x200 $S --sim-exchange-gbps 200
  JENGA_SP_OVERLAP=0 run E_s8_ov0_x300 $S --sim-exchange-gbps 300
  run E_s8_turbo_x300 $S --sim-exchange-gbps 300 --preset turbo-mgpu
  JENGA_BENCH_FORCE_DIST=1 run E_force_dist --no-cpu-baseline --no-dense-ref --steps 3
  brief $O/E_*.json
  bash tools/prof_bench.sh r04_default --no-dense-ref --no-cpu-baseline --no-secondary --no-wan-extra > $O/E_prof.log 2>&1; tail -3 $O/E_prof.log
  head -30 gpurun_out/prof_r04_default/kernel_stats.csv
  ;;
F)
  # smoke of the multi-rank launch path with one rank, wall time of the default command, the wave-per-row LayerNorm+modulate
  timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dit.py tests/test_gpu_sp_dit.py -q -m gpu -x > $O/F_tests.log 2>&1; tail -4 $O/F_tests.log
  JENGA_BENCH_FORCE_DIST=1 run F_force_dist --no-cpu-baseline --no-dense-ref --steps 3
  /usr/bin/time -v python bench.py > $O/F_default.json 2> $O/F_default.err; grep -E "Elapsed|Maximum resident" $O/F_default.err
  brief $O/F_*.json
  python - $O/F_default.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k,v in d["roofline_secondary"].items():
    if isinstance(v,dict) and "ms" in v: print(k, v["ms"], v["achieved"], v["frac"])
PY
  ;;
G)
  # wall time of the default command; the Wan-shape cross-attention test; the default record with the wave-per-row LayerNorm
  timeout 600 python -m pytest tests/test_gpu_wan_dit.py -q -m gpu -x > $O/G_tests.log 2>&1; grep -E "passed|failed" $O/G_tests.log
  T0=$(date +%s); python bench.py > $O/G_default.json 2> $O/G_default.err; T1=$(date +%s); echo "default bench wall seconds: $((T1-T0))"
  brief $O/G_default.json
  python - $O/G_default.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1])
for k,v in d["roofline_secondary"].items():
    if isinstance(v,dict) and "ms" in v: print(k, v["ms"], v["achieved"], v["frac"])
print(d["extra"]["wan14b"]["s_per_video_two_rate_estimate"], d["cpu_baseline"]["value"])
PY
  ;;
H)
  # per-rank GEMMs of the overlap path through jenga_linear (candidate timing): parity, then the three exchange rates again
  timeout 600 python -m pytest tests/test_gpu_sp_dit.py tests/test_gpu_rccl.py tests/test_gpu_dit.py -q -m gpu -x > $O/H_tests.log 2>&1; grep -E "passed|failed" $O/H_tests.log
  S="--simulate-ranks 8 --steps 6 --no-cpu-baseline --no-dense-ref"
  run H_s8_x0 $S
  run H_s8_x300 $S --sim-exchange-gbps 300
  run H_s8_x400 $S --sim-exchange-gbps 400
  run H_s8_x200 $S --sim-exchange-gbps 200
  run H_default --no-cpu-baseline --no-dense-ref --no-secondary --no-wan-extra
  brief $O/H_*.json
  ;;
Z)
  # the round's final records, one session: suite, smoke, default line, Wan line, the simulated 8-rank job at four rates,
  # program order, MLP-tail sweep, Turbo
  timeout 1500 python -m pytest tests -q -m gpu > $O/Z_suite.log 2>&1; grep -E "passed|failed" $O/Z_suite.log
  python __graft_entry__.py --smoke > $O/Z_smoke.log 2>&1; tail -1 $O/Z_smoke.log
  run Z_default
  S="--simulate-ranks 8 --steps 6 --no-cpu-baseline --no-dense-ref"
  run Z_s8_x0 $S
  run Z_s8_x300 $S --sim-exchange-gbps 300
  run Z_s8_x400 $S --sim-exchange-gbps 400
  run Z_s8_x200 $S --sim-exchange-gbps 200
  JENGA_SP_OVERLAP=0 run Z_s8_ov0_x300 $S --sim-exchange-gbps 300
  JENGA_SP_OVERLAP=0 run Z_s8_ov0_x0 $S
  JENGA_SP_MLP_TAIL=0.25 run Z_s8_tail25_x300 $S --sim-exchange-gbps 300
  JENGA_SP_MLP_TAIL=0.5 run Z_s8_tail50_x300 $S --sim-exchange-gbps 300
  run Z_s8_turbo_x300 $S --sim-exchange-gbps 300 --preset turbo-mgpu
  run Z_turbo --preset turbo --no-cpu-baseline --no-dense-ref --no-secondary --no-wan-extra
  brief $O/Z_*.json
  timeout 900 python bench.py --workload wan14b > $O/Z_wan14b.json 2> $O/Z_wan14b.err; tail -c 300 $O/Z_wan14b.json
  ;;
Y)
  # this round's counter passes of the default attention kernel (the file roofline.traffic derives from), the full 50-step
  # loop, configs[4]'s per-rank work with the replayed exchange
  bash tools/pmc_attn2.sh r04_lp --drop 0.7 --iters 2 --attn-only --flags 25 > $O/Y_pmc_lp.log 2>&1; grep -A14 '"derived"' $O/Y_pmc_lp.log | head -24
  run Y_full50 --steps 50 --no-cpu-baseline --no-dense-ref --no-secondary --no-wan-extra
  run Y_s8_3stage_i2v_x300 --simulate-ranks 8 --steps 6 --no-cpu-baseline --no-dense-ref --sim-exchange-gbps 300 --preset 3stage-mgpu --i2v
  run Y_3stage_i2v --preset 3stage --i2v --no-cpu-baseline --no-dense-ref --no-secondary --no-wan-extra
  brief $O/Y_*.json
  ;;
I)
  # host time per launch of the ctypes wrappers with and without the redundant device guard; the host-bound per-rank steps
  python tools/host_overhead.py > $O/I_host_new.json 2>/dev/null; cat $O/I_host_new.json
  JENGA_DEVICE_GUARD=always python tools/host_overhead.py > $O/I_host_old.json 2>/dev/null; cat $O/I_host_old.json
  S="--simulate-ranks 8 --steps 6 --no-cpu-baseline --no-dense-ref --sim-exchange-gbps 300"
  run I_s8_3stage_i2v_new $S --preset 3stage-mgpu --i2v
  JENGA_DEVICE_GUARD=always run I_s8_3stage_i2v_old $S --preset 3stage-mgpu --i2v
  run I_s8_base_new $S
  JENGA_DEVICE_GUARD=always run I_s8_base_old $S
  brief $O/I_*.json
  ;;
J)
  # rotated list walk (JENGA_ATTN_ROTATE = 128): parity, then the period sweep against the default order on one box
  timeout 600 python -m pytest tests/test_gpu_order.py tests/test_gpu_parity.py -q -m gpu -x > $O/J_tests.log 2>&1; grep -E "passed|failed" $O/J_tests.log; grep -E "^E " $O/J_tests.log | head -5
  ba() { tag=$1; shift; timeout 600 python tools/bench_attn.py "$@" > $O/J_attn_$tag.json 2> $O/J_attn_$tag.err; python - $O/J_attn_$tag.json $tag <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], {k: (round(d[k],3) if isinstance(d[k],float) else d[k]) for k in ("attn_ms","attn_TFLOPs","kept_mean","adjacent_shared_frac","flags","finite") if k in d})
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  }
  A="--drop 0.7 --iters 60 --attn-only"
  ba flat_base $A --flags 25
  for P in 500 700 800 875 950 1100 1500; do JENGA_ROTATE_PERIOD_US=$P ba flat_rot$P $A --flags 153; done
  ba flat_base2 $A --flags 25
  JENGA_ROTATE_PERIOD_US=875 ba coh3_rot875 $A --coherent 3 --gain 2 --flags 153
  ba coh3_base $A --coherent 3 --gain 2 --flags 25
  ;;
K)
  # rotated walk: the deterministic position mode against the clock mode and the default order; counters; the whole loop
  timeout 600 python -m pytest tests/test_gpu_order.py -q -m gpu -x > $O/K_tests.log 2>&1; grep -E "passed|failed" $O/K_tests.log; grep -E "^E " $O/K_tests.log | head -5
  ba() { tag=$1; shift; timeout 600 python tools/bench_attn.py "$@" > $O/K_attn_$tag.json 2> $O/K_attn_$tag.err; python - $O/K_attn_$tag.json $tag <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], {k: (round(d[k],3) if isinstance(d[k],float) else d[k]) for k in ("attn_ms","attn_TFLOPs","kept_mean","adjacent_shared_frac","flags","finite") if k in d})
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  }
  A="--drop 0.7 --iters 60 --attn-only"
  ba flat_base $A --flags 25
  JENGA_ROTATE_PERIOD_US=875 ba flat_clock875 $A --flags 153
  for S in 48 56 64 72 96 128; do JENGA_ROTATE_SLOTS=$S ba flat_pos$S $A --flags 153; done
  ba flat_base2 $A --flags 25
  JENGA_ROTATE_SLOTS=64 ba coh3_pos64 $A --coherent 3 --gain 2 --flags 153
  ba coh3_base $A --coherent 3 --gain 2 --flags 25
  JENGA_ROTATE_SLOTS=64 ba n8_pos64 --heads 3 --drop 0.75 --iters 200 --attn-only --flags 153
  ba n8_base --heads 3 --drop 0.75 --iters 200 --attn-only --flags 25
  JENGA_ROTATE_PERIOD_US=875 bash tools/pmc_attn2.sh r04_rot_clock --drop 0.7 --iters 2 --attn-only --flags 153 > $O/K_pmc_clock.log 2>&1; grep -A12 '"derived"' $O/K_pmc_clock.log | grep -E "per_kept_pair|l2_hit|mfma_busy|effective_clock"
  JENGA_ROTATE_SLOTS=64 bash tools/pmc_attn2.sh r04_rot_pos --drop 0.7 --iters 2 --attn-only --flags 153 > $O/K_pmc_pos.log 2>&1; grep -A12 '"derived"' $O/K_pmc_pos.log | grep -E "per_kept_pair|l2_hit|mfma_busy|effective_clock"
  B="--no-cpu-baseline --no-dense-ref --no-secondary --no-wan-extra"
  run K_default $B
  JENGA_ATTN_FLAGS=153 JENGA_ROTATE_SLOTS=64 run K_rot_pos64 $B
  JENGA_ATTN_FLAGS=153 JENGA_ROTATE_PERIOD_US=875 run K_rot_clock875 $B
  brief $O/K_default.json $O/K_rot_*.json
  ;;
L)
  # rotated walk, clock mode: automatic period (the previous launch's workgroup lifetime) vs fixed periods, pacing, the loop
  timeout 600 python -m pytest tests/test_gpu_order.py -q -m gpu -x > $O/L_tests.log 2>&1; grep -E "passed|failed" $O/L_tests.log; grep -E "^E " $O/L_tests.log | head -5
  ba() { tag=$1; shift; timeout 600 python tools/bench_attn.py "$@" > $O/L_attn_$tag.json 2> $O/L_attn_$tag.err; python - $O/L_attn_$tag.json $tag <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], {k: (round(d[k],3) if isinstance(d[k],float) else d[k]) for k in ("attn_ms","attn_TFLOPs","kept_mean","adjacent_shared_frac","flags","finite") if k in d})
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  }
  A="--drop 0.7 --iters 60 --attn-only"
  ba flat_base $A --flags 25
  ba flat_auto $A --flags 153
  JENGA_ROTATE_PERIOD_US=875 ba flat_875 $A --flags 153
  for PC in 10 25 50 100; do JENGA_ROTATE_PERIOD_US=875 JENGA_ROTATE_PACE_US=$PC ba flat_875_pace$PC $A --flags 153; done
  ba flat_base2 $A --flags 25
  ba d80_base --drop 0.8 --iters 60 --attn-only --flags 25
  ba d80_auto --drop 0.8 --iters 60 --attn-only --flags 153
  JENGA_ROTATE_PERIOD_US=875 ba d80_875 --drop 0.8 --iters 60 --attn-only --flags 153
  ba n8_base --heads 3 --drop 0.75 --iters 200 --attn-only --flags 25
  ba n8_auto --heads 3 --drop 0.75 --iters 200 --attn-only --flags 153
  ba coh3_base $A --coherent 3 --gain 2 --flags 25
  ba coh3_auto $A --coherent 3 --gain 2 --flags 153
  B="--no-cpu-baseline --no-dense-ref --no-secondary --no-wan-extra"
  run L_default $B
  JENGA_ATTN_FLAGS=153 run L_rot_auto $B
  run L_default2 $B
  JENGA_ATTN_FLAGS=153 run L_rot_auto2 $B
  brief $O/L_default*.json $O/L_rot_*.json
  ;;
M)
  # rotated walk with the mid-queue period estimate: attention A/B, then the default command with its attn_rotate leg
  timeout 600 python -m pytest tests/test_gpu_order.py -q -m gpu -x > $O/M_tests.log 2>&1; grep -E "passed|failed" $O/M_tests.log; grep -E "^E " $O/M_tests.log | head -5
  ba() { tag=$1; shift; timeout 600 python tools/bench_attn.py "$@" > $O/M_attn_$tag.json 2> $O/M_attn_$tag.err; python - $O/M_attn_$tag.json $tag <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], {k: (round(d[k],3) if isinstance(d[k],float) else d[k]) for k in ("attn_ms","attn_TFLOPs","kept_mean","adjacent_shared_frac","flags","finite") if k in d})
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  }
  A="--drop 0.7 --iters 60 --attn-only"
  ba flat_base $A --flags 25
  ba flat_auto $A --flags 153
  JENGA_ROTATE_PERIOD_US=875 ba flat_875 $A --flags 153
  ba flat_auto2 $A --flags 153
  ba flat_base2 $A --flags 25
  ba n8_base --heads 3 --drop 0.75 --iters 200 --attn-only --flags 25
  ba n8_auto --heads 3 --drop 0.75 --iters 200 --attn-only --flags 153
  bash tools/pmc_attn2.sh r04_rot_auto --drop 0.7 --iters 3 --attn-only --flags 153 > $O/M_pmc_auto.log 2>&1; grep -A12 '"derived"' $O/M_pmc_auto.log | grep -E "per_kept_pair|l2_hit|mfma_busy|effective_clock"
  run M_default --no-cpu-baseline --no-wan-extra --no-secondary
  brief $O/M_default.json
  python - $O/M_default.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1])
print(d["extra"]["attn_rotate"])
PY
  ;;
N)
  # the oracle-level parity tests with the rotated walk as the default flags (which tests hold, which bit-exactness claims
  # do not), then the final default record with the attn_rotate leg, then the full suite at HEAD
  JENGA_ATTN_FLAGS=153 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dit.py tests/test_gpu_wan_dit.py tests/test_gpu_select.py tests/test_gpu_fused.py -q -m gpu > $O/N_rotate_parity.log 2>&1; grep -E "passed|failed" $O/N_rotate_parity.log; grep -E "^FAILED" $O/N_rotate_parity.log | head -20
  JENGA_ATTN_FLAGS=153 timeout 900 python -m pytest tests/test_gpu_sp_dit.py tests/test_gpu_ulysses.py tests/test_gpu_rccl.py -q -m gpu > $O/N_rotate_sp.log 2>&1; grep -E "passed|failed" $O/N_rotate_sp.log; grep -E "^FAILED" $O/N_rotate_sp.log | head -20
  run N_default
  brief $O/N_default.json
  python - $O/N_default.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1])
print(d["extra"]["attn_rotate"]["ms_per_computed_step"], d["extra"]["attn_rotate"]["s_per_video_estimate"], d["extra"]["attn_rotate"]["attention_frac_of_peak"])
PY
  timeout 1500 python -m pytest tests -q -m gpu > $O/N_suite.log 2>&1; grep -E "passed|failed" $O/N_suite.log
  ;;
O)
  # rotated walk, deterministic position mode with the growing-spread model of the start times, against clock mode
  timeout 600 python -m pytest tests/test_gpu_order.py -q -m gpu -x > $O/O_tests.log 2>&1; grep -E "passed|failed" $O/O_tests.log; grep -E "^E " $O/O_tests.log | head -5
  ba() { tag=$1; shift; timeout 600 python tools/bench_attn.py "$@" > $O/O_attn_$tag.json 2> $O/O_attn_$tag.err; python - $O/O_attn_$tag.json $tag <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], {k: (round(d[k],3) if isinstance(d[k],float) else d[k]) for k in ("attn_ms","attn_TFLOPs","flags") if k in d})
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  }
  A="--drop 0.7 --iters 60 --attn-only"
  ba flat_base $A --flags 25
  ba flat_clock_auto $A --flags 153
  for SP in 0.0 0.15 0.3 0.42 0.6 1.0; do JENGA_ROTATE_SLOTS=64 JENGA_ROTATE_SPREAD=$SP ba flat_pos64_sp$SP $A --flags 153; done
  JENGA_ROTATE_SLOTS=58 JENGA_ROTATE_SPREAD=0.42 ba flat_pos58_sp0.42 $A --flags 153
  ba flat_base2 $A --flags 25
  ;;
P)
  # final records at HEAD: suite, smoke, the default command (with its attn_rotate and Wan legs), the Wan line in both modes
  timeout 1500 python -m pytest tests -q -m gpu > $O/P_suite.log 2>&1; grep -E "passed|failed" $O/P_suite.log
  python __graft_entry__.py --smoke > $O/P_smoke.log 2>&1; tail -1 $O/P_smoke.log
  T0=$(date +%s); python bench.py > $O/P_default.json 2> $O/P_default.err; T1=$(date +%s); echo "default bench wall seconds: $((T1-T0))"
  brief $O/P_default.json
  python - $O/P_default.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1])
print(d["extra"]["attn_rotate"]["ms_per_computed_step"], d["extra"]["attn_rotate"]["s_per_video_estimate"], d["extra"]["attn_rotate"]["attention_frac_of_peak"], d["extra"]["wan14b"]["s_per_video_two_rate_estimate"])
PY
  timeout 900 python bench.py --workload wan14b --no-cpu-baseline > $O/P_wan14b.json 2> $O/P_wan14b.err; python - $O/P_wan14b.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1])
print("wan default", d["value"], d["roofline"]["frac"])
PY
  JENGA_ATTN_FLAGS=153 timeout 900 python bench.py --workload wan14b --no-cpu-baseline > $O/P_wan14b_rotate.json 2> $O/P_wan14b_rotate.err; python - $O/P_wan14b_rotate.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1])
print("wan rotate", d["value"], d["roofline"]["frac"])
PY
  ;;
Q)
  # the full 50-step loop, measured, in both launch modes on one box
  B="--steps 50 --no-cpu-baseline --no-dense-ref --no-secondary --no-wan-extra --no-rotate-ref"
  run Q_full50_default $B
  JENGA_ATTN_FLAGS=153 run Q_full50_rotate $B
  brief $O/Q_*.json
  ;;
R)
  # rotated walk with whole heads per XCD against the default partition (variant removed after this session: the
  # JENGA_ROTATE_HEADMAP switch exists in commit 'Rotated walk with whole heads per XCD measured' only)
  timeout 600 python -m pytest tests/test_gpu_order.py -q -m gpu -x > $O/R_tests.log 2>&1; grep -E "passed|failed" $O/R_tests.log; grep -E "^E " $O/R_tests.log | head -5
  ba() { tag=$1; shift; timeout 600 python tools/bench_attn.py "$@" > $O/R_attn_$tag.json 2> $O/R_attn_$tag.err; python - $O/R_attn_$tag.json $tag <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], {k: (round(d[k],3) if isinstance(d[k],float) else d[k]) for k in ("attn_ms","attn_TFLOPs","flags") if k in d})
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  }
  A="--drop 0.7 --iters 60 --attn-only"
  ba flat_base $A --flags 25
  ba flat_rot $A --flags 153
  JENGA_ROTATE_HEADMAP=1 ba flat_rot_headmap $A --flags 153
  JENGA_ROTATE_HEADMAP=1 JENGA_ROTATE_PERIOD_US=875 ba flat_rot_headmap_875 $A --flags 153
  ba flat_rot2 $A --flags 153
  ba flat_base2 $A --flags 25
  JENGA_ROTATE_HEADMAP=1 bash tools/pmc_attn2.sh r04_rot_headmap --drop 0.7 --iters 3 --attn-only --flags 153 > $O/R_pmc.log 2>&1; grep -A12 '"derived"' $O/R_pmc.log | grep -E "per_kept_pair|l2_hit|mfma_busy|effective_clock"
  ;;
S)
  # HEAD check: suite, smoke, a short default run
  timeout 1500 python -m pytest tests -q -m gpu > $O/S_suite.log 2>&1; grep -E "passed|failed" $O/S_suite.log
  python __graft_entry__.py --smoke > $O/S_smoke.log 2>&1; tail -1 $O/S_smoke.log
  run S_default --no-cpu-baseline --no-wan-extra --no-secondary --no-dense-ref
  brief $O/S_default.json
  ;;
T)
  # rotated walk + laggard jumps against plain rotation and the default order (variant removed after these sessions: its
  # source is the commit 'EXPERIMENT (reverted by the next commit): rotated walk + laggard jumps'; profiles/r04_attn_rotate_ab.json)
  timeout 600 python -m pytest tests/test_gpu_order.py -q -m gpu -x > $O/T_tests.log 2>&1; grep -E "passed|failed" $O/T_tests.log; grep -E "^E " $O/T_tests.log | head -5
  ba() { tag=$1; shift; timeout 600 python tools/bench_attn.py "$@" > $O/T_attn_$tag.json 2> $O/T_attn_$tag.err; python - $O/T_attn_$tag.json $tag <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], {k: (round(d[k],3) if isinstance(d[k],float) else d[k]) for k in ("attn_ms","attn_TFLOPs","flags","finite") if k in d})
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  }
  A="--drop 0.7 --iters 60 --attn-only"
  ba flat_base $A --flags 25
  ba flat_rot $A --flags 153
  JENGA_ROTATE_JUMP=1 ba flat_rot_jump $A --flags 153
  JENGA_ROTATE_JUMP=1 JENGA_ROTATE_PERIOD_US=875 ba flat_rot_jump_875 $A --flags 153
  ba flat_rot2 $A --flags 153
  ba flat_base2 $A --flags 25
  JENGA_ROTATE_JUMP=1 bash tools/pmc_attn2.sh r04_rot_jump --drop 0.7 --iters 3 --attn-only --flags 153 > $O/T_pmc.log 2>&1; grep -A12 '"derived"' $O/T_pmc.log | grep -E "per_kept_pair|l2_hit|mfma_busy|effective_clock"
  ;;
U)
  # the rotated walk where every list is the whole sequence (the dense model's launches) and at sa-drop 0.8 / 0.85
  ba() { tag=$1; shift; timeout 600 python tools/bench_attn.py "$@" > $O/U_attn_$tag.json 2> $O/U_attn_$tag.err; python - $O/U_attn_$tag.json $tag <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], {k: (round(d[k],3) if isinstance(d[k],float) else d[k]) for k in ("attn_ms","attn_TFLOPs","kept_mean","flags","finite") if k in d})
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  }
  ba dense_base --drop 0.0 --p 1.0 --iters 20 --attn-only --flags 25
  ba dense_rot --drop 0.0 --p 1.0 --iters 20 --attn-only --flags 153
  ba peaky_d85_base --drop 0.85 --p 0.3 --peaky 8 --iters 100 --attn-only --flags 25
  ba peaky_d85_rot --drop 0.85 --p 0.3 --peaky 8 --iters 100 --attn-only --flags 153
  ba dense_base2 --drop 0.0 --p 1.0 --iters 20 --attn-only --flags 25
  ;;
X)
  # cross-XCD balancing (JENGA_ATTN_BALANCE): parity of the ticket kernel, same-box A/B at the kernel level (static vs
  # balanced, with and without the rotated walk), then the loop: the default line at flags 25 / 29 / 153 / 157
  timeout 900 python -m pytest tests/test_gpu_order.py -x -q -m gpu > $O/X_order.log 2>&1; tail -5 $O/X_order.log
  timeout 500 python tools/balance_ab.py --coherent --dump $O/X_times > $O/X_balance.json 2> $O/X_balance.err; cat $O/X_balance.json; tail -2 $O/X_balance.err
  L="--steps 6 --no-cpu-baseline --no-dense-ref --no-secondary --no-wan-extra --no-rotate-ref"
  JENGA_ATTN_FLAGS=25 run X_loop_25 $L
  JENGA_ATTN_FLAGS=29 run X_loop_29 $L
  JENGA_ATTN_FLAGS=157 run X_loop_157 $L
  JENGA_ATTN_FLAGS=153 run X_loop_153 $L
  JENGA_ATTN_FLAGS=29 run X_loop_29b $L
  JENGA_ATTN_FLAGS=25 run X_loop_25b $L
  brief $O/X_loop_*.json
  ;;
AA)
  # the balanced kernel (the new default): counter passes (the file roofline.traffic derives from), per-XCD finish times of
  # the static vs the balanced launch, rocprofv3 kernel stats of the default command
  bash tools/pmc_attn2.sh r04_lp_balance --drop 0.7 --iters 2 --attn-only --flags 29 > $O/AA_pmc.log 2>&1; grep -A14 '"derived"' $O/AA_pmc.log | head -24
  bash tools/prof_bench.sh r04_default --no-cpu-baseline --no-dense-ref --no-wan-extra --no-secondary > $O/AA_prof.log 2>&1; head -8 gpurun_out/prof_r04_default/kernel_stats.csv | cut -c1-200
  ;;
AB)
  # final records at the new default: suite, smoke, the default command, the 8-rank job (simulated exchange), the Wan line
  timeout 1500 python -m pytest tests -q -m gpu > $O/AB_suite.log 2>&1; grep -E "passed|failed" $O/AB_suite.log
  python __graft_entry__.py --smoke > $O/AB_smoke.log 2>&1; tail -1 $O/AB_smoke.log
  T0=$(date +%s); python bench.py > $O/AB_default.json 2> $O/AB_default.err; T1=$(date +%s); echo "default bench wall seconds: $((T1-T0))"
  S="--simulate-ranks 8 --steps 6 --no-cpu-baseline --no-dense-ref"
  run AB_s8_x0 $S
  run AB_s8_x300 $S --sim-exchange-gbps 300
  run AB_s8_x400 $S --sim-exchange-gbps 400
  brief $O/AB_default.json $O/AB_s8_*.json
  timeout 900 python bench.py --workload wan14b --no-cpu-baseline > $O/AB_wan14b.json 2> $O/AB_wan14b.err; tail -c 300 $O/AB_wan14b.json
  ;;
AE)
  # HEAD after the capture fallback: suite, smoke, the full 50-step loop and the other presets at the new default, Wan A/B
  timeout 1500 python -m pytest tests -q -m gpu > $O/AE_suite.log 2>&1; grep -E "passed|failed" $O/AE_suite.log
  python __graft_entry__.py --smoke > $O/AE_smoke.log 2>&1; tail -1 $O/AE_smoke.log
  L="--no-cpu-baseline --no-dense-ref --no-secondary --no-wan-extra"
  run AE_full50 --steps 50 $L
  run AE_turbo --preset turbo $L
  run AE_3stage_i2v --preset 3stage --i2v $L
  brief $O/AE_*.json
  for f in 25 29; do JENGA_ATTN_FLAGS=$f timeout 900 python bench.py --workload wan14b --no-cpu-baseline > $O/AE_wan14b_$f.json 2> $O/AE_wan14b_$f.err; python - $O/AE_wan14b_$f.json $f <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1])
print("wan flags", sys.argv[2], d["value"], d["roofline"]["frac"])
PY
  done
  ;;
AF)
  # hipBLASLt candidate timing: 16 (the N > 1 default) vs 32 for one rank of eight; 1 (the N = 1 default) vs 16 on one GPU
  S="--simulate-ranks 8 --steps 6 --no-cpu-baseline --no-dense-ref --no-secondary"
  L="--steps 6 --no-cpu-baseline --no-dense-ref --no-secondary --no-wan-extra --no-rotate-ref"
  run AF_s8_c16 $S
  JENGA_GEMM_CANDIDATES=32 run AF_s8_c32 $S
  run AF_one_c1 $L
  JENGA_GEMM_CANDIDATES=16 run AF_one_c16 $L
  run AF_s8_c16b $S
  brief $O/AF_*.json
  ;;
AG)
  # HEAD with the rejected launch modes moved to the experiments library: product suite, smoke, a short default line; the
  # experiment kernels once through libjenga_amd_exp.so
  timeout 1500 python -m pytest tests -q -m gpu > $O/AG_suite.log 2>&1; grep -E "passed|failed" $O/AG_suite.log
  python __graft_entry__.py --smoke > $O/AG_smoke.log 2>&1; tail -1 $O/AG_smoke.log
  JENGA_LIB=$PWD/jenga_amd/libjenga_amd_exp.so timeout 900 python -m pytest tests/test_gpu_pair.py tests/test_gpu_order.py tests/test_gpu_parity.py -q -m gpu -k "pair or sparse_kernel_vs_oracle or cohort or rotat or balanc" > $O/AG_exp.log 2>&1; tail -3 $O/AG_exp.log
  JENGA_LIB=$PWD/jenga_amd/libjenga_amd_exp.so timeout 300 python tools/rotate_replay.py --iters 20 > $O/AG_replay.json 2> $O/AG_replay.err; cat $O/AG_replay.json
  run AG_default --steps 20 --warmup 5
  brief $O/AG_default.json
  ;;
AI)
  # streaming (nontemporal) loads / stores in the row kernels: the library before (alt_libs/libjenga_amd_base.so) vs after,
  # isolated kernel rates (roofline_secondary) and the loop, interleaved; the practical copy roof of the box beside them
  ./tools/micro/hbm_copy > $O/AI_hbm_copy.txt 2>&1; grep -E "172800|D2D" $O/AI_hbm_copy.txt
  L="--steps 6 --no-cpu-baseline --no-dense-ref --no-wan-extra --no-rotate-ref"
  JENGA_LIB=$PWD/alt_libs/libjenga_amd_base.so run AI_base $L
  run AI_nt $L
  JENGA_LIB=$PWD/alt_libs/libjenga_amd_base.so run AI_base2 $L
  run AI_nt2 $L
  brief $O/AI_*.json
  python - $O/AI_base.json $O/AI_nt.json $O/AI_base2.json $O/AI_nt2.json <<'PY'
import json,sys
for f in sys.argv[1:]:
    d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith("{")][-1])
    r=d["roofline_secondary"]
    print(f.split("/")[-1], {k:(r[k]["ms"], r[k]["achieved"]) for k in ("gather_rows","ln_modulate","qk_norm_rope_pool","pack_v")})
PY
  ;;
AJ)
  # the full 50-step loop measured in both launch modes at HEAD, one box, back to back
  L="--steps 50 --no-cpu-baseline --no-dense-ref --no-secondary --no-wan-extra --no-rotate-ref"
  run AJ_full50 $L
  JENGA_ATTN_FLAGS=157 run AJ_full50_rotate $L
  brief $O/AJ_*.json
  ;;
esac
