#!/bin/bash
# one SQ pass + one GRBM pass for a bench_attn configuration: bash tools/pmc_quick.sh <tag> <bench_attn args...>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$R/gpurun_out/pmcq_$TAG
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/a -o a -- python $R/tools/bench_attn.py "$@" > $OUT/a.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_LDS -d $OUT/b -o b -- python $R/tools/bench_attn.py "$@" > $OUT/b.log 2>&1
python - <<PY
import sqlite3, glob
for db in sorted(glob.glob("$OUT/*/*.db")):
    con = sqlite3.connect(db)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    g = lambda k: [t for t in tabs if k in t][0]
    q = f"""select p.name, count(*), sum(e.value) from {g('pmc_event')} e join {g('info_pmc')} p on e.pmc_id=p.id
            join {g('kernel_dispatch')} d on e.event_id=d.event_id join {g('info_kernel_symbol')} s on d.kernel_id=s.id
            where s.kernel_name like '%bsattn%' group by p.name"""
    for r in con.execute(q):
        nd = r[1] / 32 if r[1] >= 32 else r[1]
        print(f"$TAG {r[0]:28s} {r[2] / nd:.4g}")
PY
