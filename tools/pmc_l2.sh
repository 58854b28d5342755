#!/bin/bash
# L2 hit/miss + fetch size for a bench_attn configuration: bash tools/pmc_l2.sh <tag> <bench_attn args...>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$R/gpurun_out/pmcl2_$TAG
mkdir -p $OUT
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/a -o a -- python $R/tools/bench_attn.py "$@" > $OUT/a.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/b -o b -- python $R/tools/bench_attn.py "$@" > $OUT/b.log 2>&1
python - <<PY
import sqlite3, glob
for db in sorted(glob.glob("$OUT/*/*.db")):
    con = sqlite3.connect(db)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    g = lambda k: [t for t in tabs if k in t][0]
    q = f"""select p.name, count(distinct d.id), sum(e.value) from {g('pmc_event')} e join {g('info_pmc')} p on e.pmc_id=p.id
            join {g('kernel_dispatch')} d on e.event_id=d.event_id join {g('info_kernel_symbol')} s on d.kernel_id=s.id
            where s.kernel_name like '%bsattn%' group by p.name"""
    for r in con.execute(q):
        print(f"$TAG {r[0]:28s} per-launch {r[2] / max(r[1],1):.5g}  (launches {r[1]})")
PY
