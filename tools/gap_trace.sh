#!/bin/bash
# GPU idle time inside the denoising loop: rocprofv3 --kernel-trace of a short bench run, then the gaps between
# consecutive kernels on the timeline (run on the GPU box):  bash tools/gap_trace.sh  -> gpurun_out/gap_trace/summary.json
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/gap_trace
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o t -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - $OUT <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
f = glob.glob(out + "/t/*kernel_trace.csv")[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# the timed region = from the first attention kernel of the last computed step back ... simply: the longest run of
# kernels whose gaps are < 50 ms (a step); report busy vs wall for every such run
runs, cur = [], [rows[0]]
for a, b in zip(rows, rows[1:]):
    if b[0] - a[1] > 50_000_000:
        runs.append(cur); cur = []
    cur.append(b)
runs.append(cur)
res = []
for r in runs:
    if len(r) < 200: continue
    wall = r[-1][1] - r[0][0]
    busy = sum(e - s for s, e, _ in r)
    gaps = sorted(((b[0] - a[1]) / 1e3, a[2][:60], b[2][:60]) for a, b in zip(r, r[1:]))
    big = [g for g in gaps if g[0] > 20.0]
    res.append(dict(kernels=len(r), wall_ms=wall / 1e6, busy_ms=busy / 1e6, idle_frac=1 - busy / wall,
                    gaps_over_20us=len(big), gap_us_sum_over_20us=sum(g[0] for g in big), largest=gaps[-8:]))
json.dump(res, open(out + "/summary.json", "w"), indent=1)
for x in res: print({k: v for k, v in x.items() if k != "largest"}); print(x["largest"][-4:])
PY
rm -rf $OUT/t
