#!/bin/bash
# ONE command for the first box with more than one MI355X (review item 7, round 6): RCCL has never run with N > 1 for this
# repo (no multi-GPU box was ever available to the builder), so nothing in here may be executed for the first time there.
#
#   bash tools/first_multi_gpu.sh [OUTDIR]              on the multi-GPU node: N in {2, 4, 8} up to the device count
#   DRY=1 bash tools/first_multi_gpu.sh [OUTDIR]        dry run on ONE GPU: every bench line is taken through the multi-rank code
#                                                       paths with a world of one rank (JENGA_BENCH_FORCE_DIST=1), the RCCL test
#                                                       runs its world-size-1 form; proves the script, not the fabric
#
# What it does, in order (every step logs to OUTDIR and the script goes on after a failure -- the summary says what failed):
#   1. build check (content hash of the prebuilt library against the sources)
#   2. tests/test_gpu_rccl.py: both exchange modes over real RCCL, parity against the single-rank op, exchange timing
#   3. bench.py --gpus N for the three multi-GPU presets of the reference's scripts (scripts/hyvideo_multigpu_jenga_*.sh):
#      base-mgpu, turbo-mgpu (BASELINE.json configs[2]) and 3stage-mgpu --i2v (configs[4]), N = 1 and every N > 1, in BOTH
#      exchange modes (JENGA_ULYSSES_EXCHANGE = p2p: one grouped send/recv per exchange; a2a: all_to_all_single per tensor).
#      Every line carries `roofline_xgmi` (exchange alone -> GB/s against (N-1) x 153.6 GB/s, exposed time per step, same-session
#      single-rank steps -> efficiency).
#   4. one JSON with every line and a table: preset x mode x N -> s/video, efficiency T1 / (N * TN), exposed exchange ms.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=${TMPDIR:-/tmp} HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=${1:-gpurun_out/first_multi_gpu}
mkdir -p "$OUT"
NDEV=$(python - <<'PY'
import torch
print(torch.cuda.device_count())
PY
)
if [ -n "$DRY" ]; then
  export JENGA_BENCH_FORCE_DIST=1
  NS="1"
  STEPS=${STEPS:-"--steps 6 --warmup 1"}   # (6 steps: every (stage, computed / skipped) class of a three-stage preset is sampled)
else
  NS="1"
  for n in 2 4 8; do [ "$n" -le "$NDEV" ] && NS="$NS $n"; done
  STEPS=${STEPS:-"--steps 6 --warmup 2"}
fi
echo "devices: $NDEV   N in {$NS}   dry=${DRY:-0}   out=$OUT" | tee "$OUT/summary.txt"
FAILED=""

python __graft_entry__.py > "$OUT/build.log" 2>&1 || FAILED="$FAILED build"
timeout 3600 python -m pytest tests/test_gpu_rccl.py -q -m gpu -rs > "$OUT/pytest_rccl.log" 2>&1 || FAILED="$FAILED pytest_rccl"
tail -3 "$OUT/pytest_rccl.log" | tee -a "$OUT/summary.txt"

for P in "base-mgpu" "turbo-mgpu" "3stage-mgpu --i2v"; do
  TAG=$(echo $P | tr -d ' -')
  for MODE in p2p a2a; do
    for N in $NS; do
      F="$OUT/bench_${TAG}_${MODE}_n$N"
      # (python bench.py --gpus N launches itself under torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1)
      JENGA_ULYSSES_EXCHANGE=$MODE timeout 3600 python bench.py --gpus $N --preset $P $STEPS --no-cpu-baseline --no-wan-extra \
        > "$F.json" 2> "$F.err" || FAILED="$FAILED bench_${TAG}_${MODE}_n$N"
      tail -c 300 "$F.json" | tr '\n' ' ' | cut -c1-160; echo
    done
  done
done

python - "$OUT" "$FAILED" <<'PY'
import glob, json, os, sys
out, failed = sys.argv[1], sys.argv[2].split()
lines, table = [], []
for f in sorted(glob.glob(os.path.join(out, "bench_*.json"))):
    name = os.path.basename(f)[len("bench_"):-len(".json")]
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:      # noqa: BLE001
        lines.append({"file": name, "error": repr(e)})
        continue
    d["file"] = name
    lines.append(d)
    x = d.get("roofline_xgmi") or {}
    table.append({"run": name, "n_gpus": d.get("n_gpus"), "s_per_video": d.get("value"), "ms_per_step": d.get("ms_per_step"),
                  "ranks_rccl_saw": x.get("ranks"), "exchange_mode": x.get("exchange_mode"),
                  "efficiency_same_session": x.get("efficiency"), "stages": x.get("stages")})
# efficiency from the N = 1 line of the same preset and mode: T1 / (N * TN)
by = {(t["run"].rsplit("_n", 1)[0]): t for t in table if t["n_gpus"] == 1 and not os.environ.get("JENGA_BENCH_FORCE_DIST")}
for t in table:
    base = by.get(t["run"].rsplit("_n", 1)[0])
    if base and t["n_gpus"] and t["s_per_video"]:
        t["efficiency_vs_n1_line"] = round(base["s_per_video"] / (t["n_gpus"] * t["s_per_video"]), 4)
rec = {"what": "tools/first_multi_gpu.sh: RCCL parity test + bench.py --gpus N for base-mgpu / turbo-mgpu / 3stage-mgpu --i2v in both "
               "exchange modes; dry=1 means a world of ONE rank (JENGA_BENCH_FORCE_DIST=1): the script is proven, the fabric is not",
       "dry": bool(os.environ.get("JENGA_BENCH_FORCE_DIST")), "failed_steps": failed, "table": table, "lines": lines}
json.dump(rec, open(os.path.join(out, "first_multi_gpu.json"), "w"), indent=1)
print("failed steps:", failed or "none")
for t in table:
    print(f"{t['run']:34s} N={t['n_gpus']}  {t['s_per_video']} s/video  eff(session) {t['efficiency_same_session']}  eff(vs N=1 line) {t.get('efficiency_vs_n1_line')}")
PY
