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
esac
