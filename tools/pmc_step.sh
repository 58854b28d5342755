#!/bin/bash
# HBM-side traffic of every kernel of one DiT step: rocprofv3 counter passes (no tracing domains) + one kernel-trace pass
# on `bench.py --depth 1 1 --steps 2 --warmup 1` (one double + one single block at the full 115 456-token shape).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_step
mkdir -p $OUT
CMD="python $R/bench.py --depth 1 1 --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o t -- $CMD > $OUT/t.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/f -o f -- $CMD > $OUT/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/w -o w -- $CMD > $OUT/w.log 2>&1
python - <<PY
import sqlite3, glob, csv, os
res = {}
for tag in ("f", "w"):
    db = glob.glob("$OUT/%s/*.db" % tag) + glob.glob("$OUT/%s/*/*.db" % tag)
    con = sqlite3.connect(db[0])
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    g = lambda k: [t for t in tabs if k in t][0]
    q = f"""select s.kernel_name, p.name, count(distinct d.id), sum(e.value) from {g('pmc_event')} e
            join {g('info_pmc')} p on e.pmc_id=p.id join {g('kernel_dispatch')} d on e.event_id=d.event_id
            join {g('info_kernel_symbol')} s on d.kernel_id=s.id group by s.kernel_name, p.name"""
    for name, ctr, n, tot in con.execute(q):
        res.setdefault(name, {})[ctr] = (n, tot / n)
stats = glob.glob("$OUT/t/**/t_kernel_stats.csv", recursive=True)[0]
dur = {r["Name"]: float(r["AverageNs"]) for r in csv.DictReader(open(stats))}
print("kernel,calls,avg_us,fetch_MB(2x corrected),write_MB,GBps")
for name, c in sorted(res.items(), key=lambda kv: -dur.get(kv[0], 0)):
    if "jenga" not in name and "Cijk" not in name:
        continue
    f = c.get("FETCH_SIZE", (0, 0))[1] * 2 * 1024 / 1e6
    w = c.get("WRITE_SIZE", (0, 0))[1] * 1024 / 1e6
    us = dur.get(name, 0) / 1e3
    print(f"{name[:60]},{c.get('FETCH_SIZE', (0, 0))[0]},{us:.1f},{f:.1f},{w:.1f},{(f + w) / max(us, 1e-9) * 1e3 / 1e3:.0f}")
PY
