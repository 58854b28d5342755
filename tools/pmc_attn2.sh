#!/bin/bash
# Round-2 PMC recipe for the attention kernel (run on the GPU box through gpurun):
#   bash tools/pmc_attn2.sh <tag> <tools/bench_attn.py args...>
# One --kernel-trace --stats pass, then SEPARATE --pmc passes (no tracing domains next to counters), then a summary
# JSON with the derived numbers (gpurun_out/pmc2_<tag>/summary.json; copy to profiles/ to keep).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$R/gpurun_out/pmc2_$TAG
rm -rf $OUT; mkdir -p $OUT
B="python $R/tools/bench_attn.py $*"
rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $OUT/trace -o trace -- $B > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/p1 -o p1 -- $B > $OUT/p1.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/p2 -o p2 -- $B > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/p3 -o p3 -- $B > $OUT/p3.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM -d $OUT/p4 -o p4 -- $B > $OUT/p4.log 2>&1
python - "$OUT" "$TAG" "$*" <<'PY'
import glob, json, sqlite3, sys
out, tag, args = sys.argv[1:4]
res = {"tag": tag, "command": f"bash tools/pmc_attn2.sh {tag} {args}", "counters_per_launch": {}}
for db in sorted(glob.glob(out + "/*/*.db")):
    con = sqlite3.connect(db)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    g = lambda k: [t for t in tabs if k in t][0]
    try:
        q = f"""select s.kernel_name, count(*), avg(d.end - d.start) from {g('kernel_dispatch')} d
                join {g('info_kernel_symbol')} s on d.kernel_id = s.id where s.kernel_name like '%bsattn%' group by s.kernel_name"""
        for name, n, avg in con.execute(q):
            if "/trace/" in db:
                res["kernel"] = name.split("(")[0]
                res["launches_traced"] = n
                res["kernel_avg_duration_ms_trace_pass"] = avg / 1e6
    except Exception as e:
        res.setdefault("errors", []).append(repr(e))
    try:
        q = f"""select p.name, count(distinct d.id), sum(e.value) from {g('pmc_event')} e join {g('info_pmc')} p on e.pmc_id = p.id
                join {g('kernel_dispatch')} d on e.event_id = d.event_id join {g('info_kernel_symbol')} s on d.kernel_id = s.id
                where s.kernel_name like '%bsattn%' group by p.name"""
        for name, n, tot in con.execute(q):
            res["counters_per_launch"][name] = tot / max(n, 1)
    except Exception:
        pass
for line in open(out + "/trace.log"):
    if line.startswith("{"):
        d = json.loads(line)
        res["bench"] = {k: d[k] for k in ("attn_ms", "attn_TFLOPs", "kept_mean", "pairs", "adjacent_shared_frac") if k in d}
        res["shape"] = d.get("shape")
c = res["counters_per_launch"]
der = {}
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    der["traffic_bytes_per_launch"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
    der["traffic_note"] = ("(2*FETCH_SIZE + WRITE_SIZE) KB: FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies "
                           "128-B requests at 64 B); memory-side L2 requests, Infinity-Cache hits included")
    if res.get("bench", {}).get("pairs"):
        der["kept_block_pairs_per_launch"] = res["bench"]["pairs"]
        der["traffic_bytes_per_kept_pair"] = der["traffic_bytes_per_launch"] / res["bench"]["pairs"]
if "TCC_HIT_sum" in c:
    der["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
if "GRBM_GUI_ACTIVE" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
    der["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (128.0 * c["GRBM_GUI_ACTIVE"])
    if "kernel_avg_duration_ms_trace_pass" in res:
        der["effective_clock_GHz"] = c["GRBM_GUI_ACTIVE"] / 8 / (res["kernel_avg_duration_ms_trace_pass"] * 1e6)
if "SQ_INSTS_MFMA" in c:
    der["valu_per_mfma"] = c["SQ_INSTS_VALU"] / c["SQ_INSTS_MFMA"]
    der["salu_per_mfma"] = c.get("SQ_INSTS_SALU", 0) / c["SQ_INSTS_MFMA"]
    der["lds_per_mfma"] = c.get("SQ_INSTS_LDS", 0) / c["SQ_INSTS_MFMA"]
if "SQ_WAVE_CYCLES" in c:
    w = c["SQ_WAVE_CYCLES"]
    der["wave_cycle_split"] = {"issuing": c["SQ_ACTIVE_INST_ANY"] / w, "issue_stalled": c["SQ_WAIT_INST_ANY"] / w,
                               "parked_on_waitcnt_or_barrier": c["SQ_WAIT_ANY"] / w}
res["derived"] = der
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
# keep the summaries, drop the databases (gpurun merges at most 64 MiB back)
find $OUT -name '*.db' -delete
find $OUT -name '*kernel_trace.csv' -size +2M -delete

