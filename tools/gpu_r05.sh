#!/bin/bash
# GPU sessions of round 5 (run through gpurun): bash tools/gpu_r05.sh <stage>
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r05_$1; mkdir -p $O
case "$1" in
A)  # first run of the pair kernel: parity, then LP vs pair on one box (flat lists / lists like the driver workload's), then the loop
  timeout 900 python -m pytest tests/test_gpu_pair.py tests/test_gpu_order.py -x -q -m gpu > $O/pytest_pair.log 2>&1; tail -5 $O/pytest_pair.log
  timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sparse_kernel" > $O/pytest_parity.log 2>&1; tail -3 $O/pytest_parity.log
  timeout 300 python tools/bench_attn.py --drop 0.7 --iters 40 --attn-only --flags 29 --also-flags 85 65 > $O/ab_flat.json 2> $O/ab_flat.err; tail -c 1200 $O/ab_flat.json
  timeout 300 python tools/bench_attn.py --drop 0.7 --iters 40 --attn-only --coherent 3 --gain 2 --flags 29 --also-flags 85 > $O/ab_coh.json 2> $O/ab_coh.err; tail -c 900 $O/ab_coh.json
  timeout 300 python tools/bench_attn.py --drop 0.7 --iters 40 --attn-only --pair-overlap 0.75 --flags 29 --also-flags 85 > $O/ab_ov75.json 2> $O/ab_ov75.err; tail -c 900 $O/ab_ov75.json
  timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-dense-ref --no-wan-extra > $O/bench6.json 2> $O/bench6.err; tail -c 3000 $O/bench6.json
  ;;
B)  # pair kernel v2 (V^T ring of three, work-aware order): parity, A/B, counters of both kernels on the same lists, the loop
  timeout 900 python -m pytest tests/test_gpu_pair.py tests/test_gpu_order.py -x -q -m gpu > $O/pytest_pair.log 2>&1; tail -4 $O/pytest_pair.log
  timeout 200 python tools/debug_pair.py > $O/debug.log 2>&1; grep -c "bad (qblock, head)=\[\]" $O/debug.log; grep -v "bad (qblock, head)=\[\]" $O/debug.log | head
  timeout 300 python tools/bench_attn.py --drop 0.7 --iters 40 --attn-only --flags 29 --also-flags 85 > $O/ab_flat.json 2> $O/ab_flat.err; python tools/ab_print.py $O/ab_flat.json
  timeout 300 python tools/bench_attn.py --drop 0.7 --iters 40 --attn-only --coherent 3 --gain 2 --flags 29 --also-flags 85 69 > $O/ab_coh.json 2> $O/ab_coh.err; python tools/ab_print.py $O/ab_coh.json
  timeout 300 python tools/bench_attn.py --drop 0.8 --iters 40 --attn-only --coherent 3 --gain 2 --flags 29 --also-flags 85 > $O/ab_coh08.json 2> $O/ab_coh08.err; python tools/ab_print.py $O/ab_coh08.json
  timeout 600 bash tools/pmc_attn2.sh r05_pair_coh --drop 0.7 --iters 2 --attn-only --coherent 3 --gain 2 --flags 85 > $O/pmc_pair.log 2>&1; grep -A14 '"derived"' $O/pmc_pair.log | head -24
  timeout 600 bash tools/pmc_attn2.sh r05_lp_coh --drop 0.7 --iters 2 --attn-only --coherent 3 --gain 2 --flags 29 > $O/pmc_lp.log 2>&1; grep -A14 '"derived"' $O/pmc_lp.log | head -24
  JENGA_ATTN_FLAGS=85 timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-dense-ref --no-wan-extra --no-secondary > $O/bench6_pair.json 2> $O/bench6_pair.err; python tools/ab_print.py $O/bench6_pair.json
  ;;
D)  # selection kernel + the hardened parity tests
  timeout 1200 python -m pytest tests/test_gpu_select.py -x -q -m gpu > $O/pytest_select.log 2>&1; tail -4 $O/pytest_select.log
  timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/pytest_parity.log 2>&1; tail -4 $O/pytest_parity.log
  timeout 200 python tools/bench_attn.py --drop 0.7 --iters 20 > $O/sel_flat.json 2> $O/sel_flat.err; python -c "import json;d=json.loads(open('$O/sel_flat.json').read().strip().splitlines()[-1]);print('select_ms',d['select_ms'],'pool_ms',d['pool_ms'])"
  timeout 200 python tools/bench_attn.py --drop 0.7 --iters 20 --coherent 3 --gain 2 > $O/sel_coh.json 2> $O/sel_coh.err; python -c "import json;d=json.loads(open('$O/sel_coh.json').read().strip().splitlines()[-1]);print('select_ms',d['select_ms'],'pool_ms',d['pool_ms'])"
  python __graft_entry__.py --smoke 2>&1 | tail -2
  ;;
H)  # the driver's command at HEAD (default flags), then the same command under rocprofv3 --kernel-trace --stats, then the full GPU suite
  timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python tools/ab_print.py $O/bench_default.json
  timeout 1200 bash tools/prof_bench.sh r05_default > $O/prof.log 2>&1; head -12 gpurun_out/prof_r05_default/kernel_stats.csv | cut -c1-150
  timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
  ;;
K)  # the command the way the driver runs it (--steps 20 --warmup 5), plain and under rocprofv3 without the extra legs
  timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err; python tools/ab_print.py $O/bench_steps20.json
  timeout 1200 bash tools/prof_bench.sh r05_steps20 --steps 20 --warmup 5 --no-dense-ref --no-other-kernel-ref --no-wan-extra --no-cpu-baseline > $O/prof.log 2>&1; head -8 gpurun_out/prof_r05_steps20/kernel_stats.csv | cut -c1-150; python tools/ab_print.py gpurun_out/prof_r05_steps20/bench.json | head -2
  ;;
G)  # counter passes of the default kernel at both drop rates of the Base preset, one box (roofline.traffic_per_rate)
  for R in 0.7 0.8; do
    timeout 900 bash tools/pmc_attn2.sh r05_lp_flat_$R --drop $R --iters 2 --attn-only --flags 29 > $O/pmc_flat_$R.log 2>&1; grep -E "per_kept_pair|l2_hit|mfma_busy|effective_clock" $O/pmc_flat_$R.log
    timeout 900 bash tools/pmc_attn2.sh r05_lp_coh_$R --drop $R --iters 2 --attn-only --coherent 3 --gain 2 --flags 29 > $O/pmc_coh_$R.log 2>&1; grep -E "per_kept_pair|l2_hit|mfma_busy|effective_clock" $O/pmc_coh_$R.log
  done
  ;;
F)  # sequence-parallel path: head-group pipeline parity, the xgmi record on a world of one rank, simulated 8 ranks
  timeout 1500 python -m pytest tests/test_gpu_sp_dit.py tests/test_gpu_ulysses.py tests/test_gpu_rccl.py tests/test_gpu_dit.py -x -q -m gpu > $O/pytest_sp.log 2>&1; tail -4 $O/pytest_sp.log
  JENGA_BENCH_FORCE_DIST=1 timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-dense-ref > $O/bench_force_dist.json 2> $O/bench_force_dist.err; python -c "
import json;d=json.loads(open('$O/bench_force_dist.json').read().strip().splitlines()[-1]);print('value',d['value']);print(json.dumps(d.get('roofline_xgmi'),indent=0)[:1500])"
  L="--steps 6 --warmup 2 --no-cpu-baseline --no-dense-ref --no-secondary --no-wan-extra --simulate-ranks 8"
  for G in 300 150; do
    timeout 600 python bench.py $L --sim-exchange-gbps $G > $O/sim8_x$G.json 2> $O/sim8_x$G.err; python -c "import json;d=json.loads(open('$O/sim8_x$G.json').read().strip().splitlines()[-1]);print('sim8 x$G plain    ',d['value'])"
    JENGA_ULYSSES_PIPELINE=1 timeout 600 python bench.py $L --sim-exchange-gbps $G > $O/sim8_x${G}_pipe.json 2> $O/sim8_x${G}_pipe.err; python -c "import json;d=json.loads(open('$O/sim8_x${G}_pipe.json').read().strip().splitlines()[-1]);print('sim8 x$G pipelined',d['value'])"
  done
  ;;
I)  # selection kernel, wave-per-row form: tests, the new reference-kernel goldens, clock + elimination builds
  timeout 1500 python -m pytest tests/test_gpu_select.py -x -q -m gpu > $O/pytest_select.log 2>&1; tail -4 $O/pytest_select.log
  timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "product_dtype or whole_op or randomized" > $O/pytest_parity.log 2>&1; tail -3 $O/pytest_parity.log
  for lib in ${LIBS:-base sx1 sx2 sx4 sx8 sx15}; do
    if [ "$lib" = base ]; then unset JENGA_LIB; else export JENGA_LIB=$PWD/alt_libs/$lib.so; fi
    for L in "--drop 0.7" "--drop 0.7 --coherent 3 --gain 2"; do
      timeout 100 python tools/bench_attn.py $L --iters 20 --flags 29 2> $O/$lib.err | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$lib', '$L', 'select_ms %.4f kept %.1f' % (d['select_ms'], d['kept_mean']))"
    done
  done
  unset JENGA_LIB
  ;;
E)  # selection kernel: rows per workgroup and elimination builds (clock only)
  for lib in ${LIBS:-base g1 g2 g8 sx1 sx2 sx4 sx8 sx15}; do
    if [ "$lib" = base ]; then unset JENGA_LIB; else export JENGA_LIB=$PWD/alt_libs/$lib.so; fi
    for L in "--drop 0.7" "--drop 0.7 --coherent 3 --gain 2"; do
      timeout 100 python tools/bench_attn.py $L --iters 20 --flags 29 2> $O/$lib.err | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$lib', '$L', 'select_ms %.4f kept %.1f' % (d['select_ms'], d['kept_mean']))"
    done
  done
  ;;
C)  # elimination builds of the pair kernel (wrong results, clock only): what the LDS-DMA issue and the step barrier cost
  A="--drop 0.7 --iters 30 --attn-only --coherent 3 --gain 2 --flags 85"
  for lib in ${LIBS:-base nodma nobar nodmabar}; do
    if [ "$lib" = base ]; then unset JENGA_LIB; else export JENGA_LIB=$PWD/alt_libs/$lib.so; fi
    timeout 200 python tools/bench_attn.py $A > $O/$lib.json 2> $O/$lib.err; echo $lib; python tools/ab_print.py $O/$lib.json | tail -1
  done
  ;;
esac
