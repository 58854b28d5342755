#!/bin/bash
# GPU sessions of round 6 (run through gpurun): bash tools/gpu_r06.sh <stage>
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_$1; mkdir -p $O
ab() { python tools/ab_print.py "$1"; }
case "$1" in
A)  # new parity tests, the torch-native victim test (+ kernel names under rocprofv3), the stand-alone reproducer, LP vs pair on this box
  rocm-smi --showproductname 2>/dev/null | head -8 > $O/box.txt; hostname >> $O/box.txt
  timeout 1500 python -m pytest tests/test_gpu_pool.py tests/test_gpu_ulysses.py tests/test_gpu_select.py tests/test_gpu_order.py -x -q -m gpu > $O/pytest_new.log 2>&1; tail -5 $O/pytest_new.log
  python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
  DIAG_SECS=8 timeout 300 python tools/diag_torch_victim.py > $O/torch_victim.jsonl 2> $O/torch_victim.err; cat $O/torch_victim.jsonl | cut -c1-1500
  (cd /tmp && DIAG_SECS=1 DIAG_LOADS=none timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r06_torch_victim -o tv -- python $GRAFT_REPO_ROOT/tools/diag_torch_victim.py > $GRAFT_REPO_ROOT/$O/tv_prof.log 2>&1)
  find gpurun_out/prof_r06_torch_victim -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/torch_victim_kernel_stats.csv
  bash tools/torch_pk_census.sh --intersect profiles/r06_torch_pk_selections.json.gz $O/torch_victim_kernel_stats.csv > $O/torch_victim_intersect.txt 2>&1; tail -25 $O/torch_victim_intersect.txt | cut -c1-220
  timeout 300 tools/micro/bin/pk_beside_mfma 4000 > $O/pk_beside_mfma.jsonl 2>&1; grep -v '"differing": 0' $O/pk_beside_mfma.jsonl | cut -c1-200
  timeout 300 python tools/bench_attn.py --drop 0.7 --iters 40 --attn-only --coherent 3 --gain 2 --flags 29 --also-flags 85 > $O/ab_coh.json 2> $O/ab_coh.err; ab $O/ab_coh.json
  timeout 300 python tools/bench_attn.py --drop 0.7 --iters 40 --attn-only --flags 29 --also-flags 85 > $O/ab_flat.json 2> $O/ab_flat.err; ab $O/ab_flat.json
  ;;
B)  # issue-slot diet (review item 1): alt libs of both kernels on ONE box, two interleaved passes; parity of the candidates
  C="--drop 0.7 --iters 40 --attn-only --coherent 3 --gain 2"
  for pass in 1 2; do
    for L in base p_dot2 p_m0 p_both p_w4 p_w6; do
      [ $L = base ] && unset JENGA_LIB || export JENGA_LIB=$PWD/alt_libs/$L.so
      timeout 200 python tools/bench_attn.py $C --flags 85 > $O/pair_${L}_$pass.json 2> $O/pair_${L}_$pass.err; echo "pair $L pass $pass: $(ab $O/pair_${L}_$pass.json | sed -n 2p)"
    done
    for L in base l_dot2 l_m0 l_both; do
      [ $L = base ] && unset JENGA_LIB || export JENGA_LIB=$PWD/alt_libs/$L.so
      timeout 200 python tools/bench_attn.py $C --flags 29 > $O/lp_${L}_$pass.json 2> $O/lp_${L}_$pass.err; echo "LP   $L pass $pass: $(ab $O/lp_${L}_$pass.json | sed -n 2p)"
    done
  done
  for L in p_w4 p_w6 l_both; do
    JENGA_LIB=$PWD/alt_libs/$L.so timeout 900 python -m pytest tests/test_gpu_pair.py tests/test_gpu_order.py "tests/test_gpu_parity.py" -x -q -m gpu -k "pair or order or full_size or sparse_kernel or narrow or whole_op or running_max" > $O/pytest_$L.log 2>&1; echo "parity $L: $(tail -1 $O/pytest_$L.log)"
  done
  unset JENGA_LIB
  ;;
C)  # diet without dot2 (clean A/B + counters), torch victims with the broadcast-high selection, row-kernel overlap A/B in the loop
  C="--drop 0.7 --iters 40 --attn-only --coherent 3 --gain 2"
  for pass in 1 2; do
    for L in base p_m0 p_m0w4 p_m0w6; do
      [ $L = base ] && unset JENGA_LIB || export JENGA_LIB=$PWD/alt_libs/$L.so
      timeout 200 python tools/bench_attn.py $C --flags 85 > $O/pair_${L}_$pass.json 2> $O/pair_${L}_$pass.err; echo "pair $L pass $pass: $(ab $O/pair_${L}_$pass.json | sed -n 2p)"
    done
  done
  unset JENGA_LIB
  timeout 600 bash tools/pmc_attn2.sh r06_pair_base --drop 0.7 --iters 2 --attn-only --coherent 3 --gain 2 --flags 85 > $O/pmc_pair_base.log 2>&1; grep -E "valu_per_mfma|salu_per_mfma|lds_per_mfma|mfma_busy|effective_clock|issuing|issue_stalled|parked" $O/pmc_pair_base.log
  JENGA_LIB=$PWD/alt_libs/p_m0w4.so timeout 600 bash tools/pmc_attn2.sh r06_pair_m0w4 --drop 0.7 --iters 2 --attn-only --coherent 3 --gain 2 --flags 85 > $O/pmc_pair_m0w4.log 2>&1; grep -E "valu_per_mfma|salu_per_mfma|lds_per_mfma|mfma_busy|effective_clock|issuing|issue_stalled|parked" $O/pmc_pair_m0w4.log
  DIAG_SECS=8 timeout 400 python tools/diag_torch_victim.py > $O/torch_victim.jsonl 2> $O/torch_victim.err; python - <<PY
import json
for l in open("$O/torch_victim.jsonl"):
    d = json.loads(l)
    if "skipped" in d: print("skipped", d); continue
    print(d["load"], "runs", d["runs_per_op"], "load_it", d["load_iterations"], "bcast:", {k[:28]: v for k, v in d["mismatches_broadcast_high_kernels"].items()}, "swap-only bad:", sum(d["mismatches_swap_only_kernels"].values()), "control bad:", sum(d["mismatches_control"].values()))
PY
  (cd /tmp && DIAG_SECS=1 DIAG_LOADS=none timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r06_torch_victim2 -o tv -- python $GRAFT_REPO_ROOT/tools/diag_torch_victim.py > $GRAFT_REPO_ROOT/$O/tv_prof.log 2>&1)
  find gpurun_out/prof_r06_torch_victim2 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/torch_victim_kernel_stats.csv
  bash tools/torch_pk_census.sh --intersect profiles/r06_torch_pk_selections.json.gz $O/torch_victim_kernel_stats.csv > $O/torch_victim_intersect_bcast.txt 2>&1; tail -20 $O/torch_victim_intersect_bcast.txt | cut -c1-200
  timeout 400 tools/micro/bin/pk_beside_mfma 4000 > $O/pk_beside_mfma.jsonl 2>&1; grep -v '"differing": 0' $O/pk_beside_mfma.jsonl | cut -c1-220; grep -c '"differing": 0' $O/pk_beside_mfma.jsonl
  timeout 900 python -m pytest tests/test_gpu_dit.py tests/test_gpu_sp_dit.py -x -q -m gpu > $O/pytest_dit.log 2>&1; tail -3 $O/pytest_dit.log
  LB="--steps 6 --warmup 2 --no-cpu-baseline --no-dense-ref --no-wan-extra --no-secondary --no-other-kernel-ref"
  for pass in 1 2; do
    for V in 0 1; do
      JENGA_ROWOPS_OVERLAP=$V timeout 600 python bench.py $LB > $O/bench_overlap${V}_$pass.json 2> $O/bench_overlap${V}_$pass.err; echo "overlap=$V pass $pass: $(ab $O/bench_overlap${V}_$pass.json | head -1)"
    done
  done
  ;;
D)  # the in-run counter passes in a bench line, the dry run of tools/first_multi_gpu.sh, the whole GPU suite
  timeout 1200 python bench.py --steps 6 --warmup 2 --no-dense-ref --no-wan-extra --no-cpu-baseline --no-other-kernel-ref > $O/bench_pmc.json 2> $O/bench_pmc.err; ab $O/bench_pmc.json
  python - <<PY
import json
d = json.loads(open("$O/bench_pmc.json").read().strip().splitlines()[-1])["roofline"]
print(d["traffic_provenance"][:400]); print("traffic", d["traffic"], d["traffic_TBps"], d["traffic_ratio_to_algorithmic_bytes"])
for k, v in d["traffic_per_rate"].items(): print(k, {kk: v[kk] for kk in v if kk not in ("pmc_points",)})
PY
  DRY=1 timeout 2400 bash tools/first_multi_gpu.sh $O/first_multi_gpu_dry > $O/first_multi_gpu_dry.log 2>&1; tail -12 $O/first_multi_gpu_dry.log
  timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3
  ;;
E0)  # the split bench.py end to end with every leg, short: a lost name would otherwise cost the long session
  timeout 900 python bench.py --steps 2 --warmup 1 > $O/bench_short.json 2> $O/bench_short.err
  timeout 900 python bench.py --workload wan14b --steps 2 --warmup 1 > $O/wan_short.json 2> $O/wan_short.err
  python - <<PY || exit 1
import json, sys
ok = True
for f in ("$O/bench_short.json", "$O/wan_short.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["value"], "extras_failed:", d.get("extras_failed"), "keys:", sorted(k for k in d if k not in ("config",)))
        ok &= not d.get("extras_failed")
    except Exception as e:
        print(f, "NO LINE", repr(e)); ok = False
        print(open(f.replace(".json", ".err")).read()[-1500:])
sys.exit(0 if ok else 1)
PY
  ;;
E)  # the records at HEAD (review item 5): the driver's command, the full 50-step loop, Wan2.1-14B, the shipped rates -- each plain and under rocprofv3 --stats
  timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err; ab $O/bench_steps20.json
  timeout 1200 bash tools/prof_bench.sh r06_steps20 --steps 20 --warmup 5 --no-dense-ref --no-other-kernel-ref --no-wan-extra --no-cpu-baseline --no-secondary --pmc off > $O/prof_steps20.log 2>&1; head -6 gpurun_out/prof_r06_steps20/kernel_stats.csv | cut -c1-160; ab gpurun_out/prof_r06_steps20/bench.json | head -1
  timeout 1500 python bench.py --steps 50 --warmup 2 --no-wan-extra --no-cpu-baseline > $O/bench_full50.json 2> $O/bench_full50.err; ab $O/bench_full50.json
  timeout 1200 bash tools/prof_bench.sh r06_full50 --steps 50 --warmup 2 --no-dense-ref --no-other-kernel-ref --no-wan-extra --no-cpu-baseline --no-secondary --pmc off > $O/prof_full50.log 2>&1; head -6 gpurun_out/prof_r06_full50/kernel_stats.csv | cut -c1-160; ab gpurun_out/prof_r06_full50/bench.json | head -1
  timeout 1500 python bench.py --preset base-mgpu --steps 50 --warmup 2 --no-wan-extra --no-cpu-baseline > $O/bench_base_mgpu_full50.json 2> $O/bench_base_mgpu_full50.err; ab $O/bench_base_mgpu_full50.json
  timeout 2400 python bench.py --workload wan14b --steps 50 --warmup 1 > $O/bench_wan14b.json 2> $O/bench_wan14b.err; python -c "
import json;d=json.loads(open('$O/bench_wan14b.json').read().strip().splitlines()[-1]);print('wan14b', d['value'], d['unit'], d['roofline']['frac'], d['config'].get('schedule','')[:80])"
  timeout 2400 bash tools/prof_bench.sh r06_wan14b --workload wan14b --steps 50 --warmup 1 --no-cpu-baseline > $O/prof_wan14b.log 2>&1; head -6 gpurun_out/prof_r06_wan14b/kernel_stats.csv | cut -c1-160
  ;;
F)  # final: the whole GPU suite at HEAD, smoke, the default bench command, the torch victims on one more box
  timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3
  python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
  timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; ab $O/bench_default.json; python -c "
import json;d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]);print('extras_failed', d.get('extras_failed'), 'traffic', d['roofline']['traffic_TBps'], d['roofline']['traffic_provenance'][:30], 'dense', d['dense_reference']['s_per_video'])"
  DIAG_SECS=8 timeout 400 python tools/diag_torch_victim.py > $O/torch_victim.jsonl 2> $O/torch_victim.err; python - <<PY
import json
for l in open("$O/torch_victim.jsonl"):
    d = json.loads(l)
    if "skipped" in d: print("skipped", d); continue
    print(d["load"], "runs", d["runs_per_op"], "bcast bad:", {k[:28]: v for k, v in d["mismatches_broadcast_high_kernels"].items() if v}, "swap-only bad:", sum(d["mismatches_swap_only_kernels"].values()), "control bad:", sum(d["mismatches_control"].values()))
PY
  timeout 400 tools/micro/bin/pk_beside_mfma 4000 > $O/pk_beside_mfma.jsonl 2>&1; grep -v '"differing": 0' $O/pk_beside_mfma.jsonl | cut -c1-160
  ;;
G)  # fused norm / RoPE / pool kernel: Q and K in different workgroups (more waves, more loads in flight) against the one-thread-both form
  for pass in 1 2 3; do
    for L in base q_old q_ring3; do
      [ $L = base ] && unset JENGA_LIB || export JENGA_LIB=$PWD/alt_libs/$L.so
      python tools/bench_rowops.py >> $O/rowops.jsonl 2>> $O/rowops.err; tail -1 $O/rowops.jsonl
    done
  done
  unset JENGA_LIB
  timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_pool.py tests/test_gpu_parity.py -x -q -m gpu -k "fused or pool or norm or rope or whole_op" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
  ;;
H)  # the dry run of the first multi-GPU session once more, with the default six steps (every class of the three-stage preset sampled)
  DRY=1 timeout 3000 bash tools/first_multi_gpu.sh $O/first_multi_gpu_dry > $O/first_multi_gpu_dry.log 2>&1; tail -12 $O/first_multi_gpu_dry.log
  ;;
I)  # Wan blocks: gate + residual in the GEMM epilogue (fp32 C / D) -- tests, then the Wan2.1-14B loop with and without, one box
  timeout 1500 python -m pytest tests/test_gpu_fused.py tests/test_gpu_wan_dit.py tests/test_gpu_parity.py -x -q -m gpu -k "linear or wan or dense or other_forms" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
  for pass in 1 2; do
    for V in 0 1; do
      JENGA_WAN_FUSE_GATE=$V timeout 900 python bench.py --workload wan14b --steps 6 --warmup 1 --no-cpu-baseline > $O/wan_fuse${V}_$pass.json 2> $O/wan_fuse${V}_$pass.err
      python -c "
import json;d=json.loads(open('$O/wan_fuse${V}_$pass.json').read().strip().splitlines()[-1]);print('fuse=$V pass $pass', d['value'], d['roofline']['frac'], d['config'].get('ms_per_step_by_drop_rate'))"
    done
  done
  ;;
esac
