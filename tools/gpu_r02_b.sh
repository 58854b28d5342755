#!/bin/bash
# round 2, GPU call B: pair kernel with in-stream DMA; quick loop: tests + micro-bench
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r02b
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_pair.py -x -q 2>&1 | tail -25 > $O/pair_tests.log
tail -5 $O/pair_tests.log
for fl in 5 1; do
  timeout 300 python tools/bench_attn.py --iters 3 --drop 0.7 --flags $fl > $O/attn_flat_f$fl.json 2> $O/attn_flat_f$fl.err
  timeout 300 python tools/bench_attn.py --iters 3 --drop 0.7 --flags $fl --pair-overlap 0.8 > $O/attn_ov80_f$fl.json 2> $O/attn_ov80_f$fl.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02b/attn_*.json")):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], "ms=%.2f TF=%.0f frac=%.3f kept=%.0f shared=%s" % (d["attn_ms"], d["attn_TFLOPs"], d["attn_frac_of_2.5PF"], d["kept_mean"], d.get("pair_shared_frac")))
    except Exception as e: print(f, "ERR", e)
PY
