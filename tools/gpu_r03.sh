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
B)
  # second session: the whole suite at the new ABI, the experiment kernels through libjenga_amd_exp.so, the coherent-list
  # regime (LP in remap / plain / sorted order, LP pair), the N=8-shape launch, the per-rank step with the fused prologue
  timeout 1500 python -m pytest tests -q -m gpu -x > $O/B_suite.log 2>&1; tail -15 $O/B_suite.log
  JENGA_LIB=$PWD/jenga_amd/libjenga_amd_exp.so timeout 900 python -m pytest tests/test_gpu_pair.py tests/test_gpu_parity.py -q -m gpu -k "pair or sparse_kernel_vs_oracle" > $O/B_exp.log 2>&1; tail -5 $O/B_exp.log
  ba() { tag=$1; shift; timeout 300 python tools/bench_attn.py "$@" > $O/B_attn_$tag.json 2> $O/B_attn_$tag.err; python - $O/B_attn_$tag.json $tag <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], {k: (round(d[k],3) if isinstance(d[k],float) else d[k]) for k in ("attn_ms","attn_TFLOPs","kept_mean","adjacent_shared_frac","flags","finite") if k in d})
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  }
  ba flat_lp --drop 0.7 --iters 20
  ba flat_sorted --drop 0.7 --iters 20 --sorted
  ba coh3_lp --drop 0.7 --iters 20 --coherent 3 --gain 2
  ba coh3_plain --drop 0.7 --iters 20 --coherent 3 --gain 2 --flags 8
  ba coh3_sorted --drop 0.7 --iters 20 --coherent 3 --gain 2 --sorted
  ba coh8_lp --drop 0.7 --iters 20 --coherent 8 --gain 2
  ba coh3_r1 --drop 0.7 --iters 20 --coherent 3 --gain 2 --flags 1
  JENGA_LIB=$PWD/jenga_amd/libjenga_amd_exp.so ba coh3_lppair --drop 0.7 --iters 20 --coherent 3 --gain 2 --flags 73
  JENGA_LIB=$PWD/jenga_amd/libjenga_amd_exp.so ba coh8_lppair --drop 0.7 --iters 20 --coherent 8 --gain 2 --flags 73
  JENGA_LIB=$PWD/jenga_amd/libjenga_amd_exp.so ba flat_lppair --drop 0.7 --iters 20 --flags 73
  ba n8_flat --heads 3 --drop 0.75 --iters 100
  ba n8_flat_sorted --heads 3 --drop 0.75 --iters 100 --sorted
  ba n8_coh3 --heads 3 --drop 0.75 --iters 100 --coherent 3 --gain 2
  ba n8_coh3_sorted --heads 3 --drop 0.75 --iters 100 --coherent 3 --gain 2 --sorted
  run B_sim8 --simulate-ranks 8 --steps 3 --no-cpu-baseline --no-dense-ref
  run B_coherent --coherent 4 --peaky 3 --no-cpu-baseline
  ;;
C)
  # third session: counters for the coherent and the flat regime (separate --pmc passes), sustained A/B of the
  # kept-count-aware order, GEMMs at the per-rank shapes of an 8-rank job (default pick vs TunableOp), profile of the
  # per-rank step after the prologue fusion, coherent latents through the whole DiT
  bash tools/pmc_attn2.sh r03_coh3 --drop 0.7 --iters 2 --coherent 3 --gain 2 > $O/C_pmc_coh3.log 2>&1; tail -30 $O/C_pmc_coh3.log
  bash tools/pmc_attn2.sh r03_flat --drop 0.7 --iters 2 > $O/C_pmc_flat.log 2>&1; tail -30 $O/C_pmc_flat.log
  ba() { tag=$1; shift; timeout 300 python tools/bench_attn.py "$@" > $O/C_attn_$tag.json 2> $O/C_attn_$tag.err; python - $O/C_attn_$tag.json $tag <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], {k: (round(d[k],3) if isinstance(d[k],float) else d[k]) for k in ("attn_ms","attn_TFLOPs","kept_mean","adjacent_shared_frac","flags","finite") if k in d})
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  }
  ba ab_flat_plain_1 --drop 0.7 --iters 200
  ba ab_flat_sorted_1 --drop 0.7 --iters 200 --sorted
  ba ab_flat_plain_2 --drop 0.7 --iters 200
  ba ab_flat_sorted_2 --drop 0.7 --iters 200 --sorted
  ba ab_coh_plain --drop 0.7 --iters 200 --coherent 3 --gain 2
  ba ab_coh_sorted --drop 0.7 --iters 200 --coherent 3 --gain 2 --sorted
  ba ab_n8_plain --heads 3 --drop 0.75 --iters 1000
  ba ab_n8_sorted --heads 3 --drop 0.75 --iters 1000 --sorted
  timeout 300 python tools/tune_gemm.py --ranks=8 > $O/C_gemm_n8_default.txt 2>&1; cat $O/C_gemm_n8_default.txt
  PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=$O/tunableop_n8.csv timeout 900 python tools/tune_gemm.py --ranks=8 > $O/C_gemm_n8_tuned.txt 2>&1; cat $O/C_gemm_n8_tuned.txt; cat $O/tunableop_n8*.csv | head -20
  bash tools/prof_bench.sh r03_sim8_fused --simulate-ranks 8 --steps 3 --no-cpu-baseline --no-dense-ref > $O/C_prof_sim8.log 2>&1
  run C_coherent4 --coherent 4 --no-cpu-baseline --no-dense-ref
  run C_coherent4_peaky15 --coherent 4 --peaky 1.5 --no-cpu-baseline --no-dense-ref
  ;;
D)
  # fourth session: launch order variants (is it head-of-line blocking between XCDs?), TunableOp recording for the
  # GEMM shapes of the N=1 loop and of one rank of an 8-rank job, then the bench with and without the recorded picks
  ba() { tag=$1; shift; timeout 300 python tools/bench_attn.py "$@" > $O/D_attn_$tag.json 2> $O/D_attn_$tag.err; python - $O/D_attn_$tag.json $tag <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], {k: (round(d[k],3) if isinstance(d[k],float) else d[k]) for k in ("attn_ms","attn_TFLOPs","kept_mean","kept_min_max","adjacent_shared_frac","flags","finite") if k in d})
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  }
  ba flat_9 --drop 0.7 --iters 200 --flags 9
  ba flat_24 --drop 0.7 --iters 200 --flags 24
  ba flat_25 --drop 0.7 --iters 200 --flags 25
  ba flat80_9 --drop 0.8 --iters 200 --flags 9
  ba flat80_25 --drop 0.8 --iters 200 --flags 25
  rm -f $O/tuned_n1.csv $O/tuned_sim8.csv
  timeout 1200 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-dense-ref --gemm-tuning record:$O/tuned_n1.csv > $O/D_record_n1.json 2> $O/D_record_n1.err; tail -c 300 $O/D_record_n1.json; wc -l $O/tuned_n1.csv
  timeout 900 python bench.py --simulate-ranks 8 --steps 1 --warmup 0 --no-cpu-baseline --no-dense-ref --gemm-tuning record:$O/tuned_sim8.csv > $O/D_record_sim8.json 2> $O/D_record_sim8.err; wc -l $O/tuned_sim8.csv
  cat $O/tuned_n1.csv
  run D_n1_default --no-cpu-baseline --gemm-tuning off
  run D_n1_tuned --no-cpu-baseline --gemm-tuning $O/tuned_n1.csv
  run D_sim8_default --simulate-ranks 8 --steps 3 --no-cpu-baseline --no-dense-ref --gemm-tuning off
  run D_sim8_tuned --simulate-ranks 8 --steps 3 --no-cpu-baseline --no-dense-ref --gemm-tuning $O/tuned_sim8.csv
  ;;
E1)
  # counters for the default kernel + order (flat and coherent), the presets, the per-rank steps, Wan2.1-14B
  timeout 1500 python -m pytest tests -q -m gpu > $O/E_suite.log 2>&1; tail -4 $O/E_suite.log
  bash tools/pmc_attn2.sh r03_lp --drop 0.7 --iters 2 > $O/E_pmc_lp.log 2>&1; tail -22 $O/E_pmc_lp.log
  bash tools/pmc_attn2.sh r03_lp_unsorted --drop 0.7 --iters 2 --flags 9 > $O/E_pmc_lp9.log 2>&1; tail -22 $O/E_pmc_lp9.log
  run E_turbo --preset turbo --no-cpu-baseline --no-dense-ref
  run E_flash --preset flash --no-cpu-baseline --no-dense-ref
  run E_3stage --preset 3stage --no-cpu-baseline --no-dense-ref
  run E_3stage_i2v --preset 3stage --i2v --no-cpu-baseline --no-dense-ref
  run E_dense --preset dense --steps 2 --warmup 1 --no-cpu-baseline
  run E_base_shipped_rates --preset base-mgpu --no-cpu-baseline --no-dense-ref
  run E_sim8_base --simulate-ranks 8 --steps 3 --no-cpu-baseline
  run E_sim8_turbo --simulate-ranks 8 --preset turbo-mgpu --steps 3 --no-cpu-baseline --no-dense-ref
  run E_sim8_3stage_i2v --simulate-ranks 8 --preset 3stage-mgpu --i2v --steps 3 --no-cpu-baseline --no-dense-ref
  timeout 600 python tools/bench_wan.py --qk-gain 4 > $O/E_wan14b_gain4.json 2> $O/E_wan14b_gain4.err; tail -c 600 $O/E_wan14b_gain4.json
  ;;
E2)
  # closing records at HEAD: rocprofv3 kernel stats of the default command, the default bench line, the full loop
  bash tools/prof_bench.sh r03_default > $O/E2_prof.log 2>&1; head -12 gpurun_out/prof_r03_default/kernel_stats.csv | cut -c1-160
  run E2_default
  run E2_full50 --steps 50 --warmup 1 --no-cpu-baseline --no-dense-ref
  run E2_coherent --coherent 4 --no-cpu-baseline --no-dense-ref
  ;;
F)
  # A/B: the two workgroups of a CU given a start offset (alt_libs/dph*.so, tools/build_alt3.sh -DJENGA_LP_DEPHASE=n)
  bash tools/gpu_ab.sh r03/F_ab "--drop 0.7 --iters 200" base dph1 dph2 dph4 base dph2
  bash tools/gpu_ab.sh r03/F_ab_coh "--drop 0.7 --iters 200 --coherent 3 --gain 2" base dph2
  ;;
G)
  # jenga_linear (GEMM epilogues): tests, then the loop with it
  timeout 1500 python -m pytest tests -q -m gpu > $O/G_suite.log 2>&1; tail -12 $O/G_suite.log
  [ "$2" = tests ] && exit 0
  run G_default --no-cpu-baseline --no-dense-ref
  run G_sim8 --simulate-ranks 8 --steps 3 --no-cpu-baseline --no-dense-ref
  ;;
I)
  # A/B in the loop: gate + residual in the GEMM epilogue vs the separate kernel; candidate timing at the 8-rank shapes
  JENGA_FUSE_GATE=0 run I_n1_gate_unfused --no-cpu-baseline --no-dense-ref --steps 4
  JENGA_FUSE_GATE=1 run I_n1_gate_fused --no-cpu-baseline --no-dense-ref --steps 4
  JENGA_FUSE_GATE=0 run I_sim8_gate_unfused --simulate-ranks 8 --steps 3 --no-cpu-baseline --no-dense-ref
  JENGA_FUSE_GATE=1 run I_sim8_gate_fused --simulate-ranks 8 --steps 3 --no-cpu-baseline --no-dense-ref
  JENGA_FUSE_GATE=1 JENGA_GEMM_CANDIDATES=16 run I_sim8_gate_fused_k16 --simulate-ranks 8 --steps 3 --warmup 2 --no-cpu-baseline --no-dense-ref
  ;;
J)
  # same-box A/B/C of the GEMM-epilogue changes (boxes differ by ~2 %: nothing else is comparable)
  for rep in 1 2; do
    JENGA_SPLIT_LINEAR1=0 JENGA_FUSE_GATE=0 run J_n1_00_$rep --no-cpu-baseline --no-dense-ref --steps 4
    JENGA_SPLIT_LINEAR1=1 JENGA_FUSE_GATE=0 run J_n1_10_$rep --no-cpu-baseline --no-dense-ref --steps 4
    JENGA_SPLIT_LINEAR1=1 JENGA_FUSE_GATE=1 run J_n1_11_$rep --no-cpu-baseline --no-dense-ref --steps 4
  done
  JENGA_SPLIT_LINEAR1=0 JENGA_FUSE_GATE=0 run J_sim8_00 --simulate-ranks 8 --steps 3 --no-cpu-baseline --no-dense-ref
  JENGA_SPLIT_LINEAR1=1 JENGA_FUSE_GATE=0 run J_sim8_10 --simulate-ranks 8 --steps 3 --no-cpu-baseline --no-dense-ref
  ;;
L)
  # closing records at HEAD
  timeout 1500 python -m pytest tests -q -m gpu > $O/L_suite.log 2>&1; tail -3 $O/L_suite.log
  bash tools/prof_bench.sh r03_default --no-cpu-baseline --no-dense-ref > $O/L_prof.log 2>&1; head -14 gpurun_out/prof_r03_default/kernel_stats.csv | cut -c1-170
  run L_default
  run L_full50 --steps 50 --warmup 1 --no-cpu-baseline --no-dense-ref
  run L_sim8_base --simulate-ranks 8 --steps 3 --no-cpu-baseline
  run L_sim8_turbo --simulate-ranks 8 --preset turbo-mgpu --steps 3 --no-cpu-baseline --no-dense-ref
  run L_turbo --preset turbo --no-cpu-baseline --no-dense-ref
  run L_3stage --preset 3stage --no-cpu-baseline --no-dense-ref
  ;;
W)
  # Wan path: fused q/k prologue + cross-attention q cast + ffn GELU epilogue: tests, then the 14B forward
  timeout 900 python -m pytest tests/test_gpu_wan_dit.py tests/test_gpu_dit.py -q -m gpu > $O/W_tests.log 2>&1; tail -5 $O/W_tests.log
  timeout 600 python tools/bench_wan.py --qk-gain 4 > $O/W_wan14b_gain4.json 2> $O/W_wan14b_gain4.err; tail -c 500 $O/W_wan14b_gain4.json
  ;;
Z)
  # final: the whole suite at HEAD (product library, then the experiment kernels), Wan2.1-14B record + profile
  timeout 1500 python -m pytest tests -q -m gpu > $O/Z_suite.log 2>&1; tail -3 $O/Z_suite.log
  JENGA_LIB=$PWD/jenga_amd/libjenga_amd_exp.so timeout 900 python -m pytest tests/test_gpu_pair.py tests/test_gpu_parity.py -q -m gpu -k "pair or sparse_kernel_vs_oracle" > $O/Z_exp.log 2>&1; tail -2 $O/Z_exp.log
  timeout 600 python tools/bench_wan.py --qk-gain 4 > $O/Z_wan14b_gain4.json 2> $O/Z_wan14b_gain4.err; tail -c 500 $O/Z_wan14b_gain4.json
  timeout 600 python tools/bench_wan.py --task t2v-1.3B --size 832x480 --qk-gain 4 > $O/Z_wan1p3b.json 2> $O/Z_wan1p3b.err; tail -c 400 $O/Z_wan1p3b.json
  bash tools/prof_wan.sh r03_wan8 --qk-gain 4 --layers 8 > $O/Z_prof_wan.log 2>&1
  python __graft_entry__.py --smoke 2>&1 | tail -1
  ;;
esac
