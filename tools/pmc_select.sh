#!/bin/bash
# two counter passes of the selection kernel at the 720p shape: bash tools/pmc_select.sh <tag> [bench_attn args]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$R/gpurun_out/pmcsel_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES -d $OUT/a -o a -- python $R/tools/bench_attn.py --iters 3 "$@" > $OUT/a.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS -d $OUT/b -o b -- python $R/tools/bench_attn.py --iters 3 "$@" > $OUT/b.log 2>&1
python - <<PY
import sqlite3, glob, json
res = {}
for db in sorted(glob.glob("$OUT/*/*.db")):
    con = sqlite3.connect(db)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    g = lambda k: [t for t in tabs if k in t][0]
    q = f"""select p.name, count(distinct d.dispatch_id), sum(e.value) from {g('pmc_event')} e join {g('info_pmc')} p on e.pmc_id=p.id
            join {g('kernel_dispatch')} d on e.event_id=d.event_id join {g('info_kernel_symbol')} s on d.kernel_id=s.id
            where s.kernel_name like '%block_select%' group by p.name"""
    for r in con.execute(q):
        res[r[0]] = {"per_launch": r[2] / max(r[1], 1), "launches": r[1]}
print(json.dumps(res))
PY
