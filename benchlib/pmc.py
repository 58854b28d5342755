"""Part of bench.py (repo root): roofline.traffic -- the committed per-pair constants and the counter passes of the run itself.  Split out of bench.py in round 6;
bench.py re-exports these names."""
import json
import os
import sys
import time

from .consts import ATTN_ALGORITHMIC_BYTES, ROOT


def traffic_bytes_per_pair(pmc, rate, shared_frac):
    """Memory-side bytes per kept block pair of the LP kernel at drop rate `rate`, from the committed counter passes
    (profiles/PMC_FILE): the passes of the nearest measured rate, interpolated linearly in adjacent_shared_frac between its
    'flat' and 'coh' points (clamped to them).  -> dict(bytes_per_pair, rate, points) or None."""
    rates = pmc.get("rates") or {}
    if not rates or rate is None:
        return None
    key = min(rates, key=lambda r: abs(float(r) - float(rate)))
    flat, coh = rates[key]["flat"], rates[key]["coh"]
    f0, f1 = flat["adjacent_shared_frac"], coh["adjacent_shared_frac"]
    x = f0 if shared_frac is None or shared_frac != shared_frac else min(max(shared_frac, f0), f1)
    w = (x - f0) / max(f1 - f0, 1e-9)
    b = flat["traffic_bytes_per_kept_pair"] * (1 - w) + coh["traffic_bytes_per_kept_pair"] * w
    return {"bytes_per_pair": b, "rate": float(key),
            "points": {"flat": [round(f0, 3), round(flat["traffic_bytes_per_kept_pair"])],
                       "coh": [round(f1, 3), round(coh["traffic_bytes_per_kept_pair"])], "at_shared_frac": round(x, 3)}}


def pmc_read_in_this_run(argv, timeout_s=360):
    """-> ({tag: {"fetch_bytes", "write_bytes", "pairs", "launches"}}, "ok") or (None, why).  Two child processes of this
    script (`--pmc-child`: the same model, seeds, lists; one computed step per stage) under rocprofv3 --pmc, one counter set
    each (FETCH_SIZE needs 3 of the 4 TCC counters, WRITE_SIZE 2: MI355X_MICROARCH.md), counters only.  Bytes with the guide's
    gfx950 corrections: FETCH_SIZE counts 64 B per 128-B request -> x 2; both are in KiB."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        return None, "rocprofv3 not found"
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        # this process is itself being profiled (`rocprofv3 ... -- python bench.py`): a profiler inside a profiler is not a
        # configuration anybody tests, and the guide forbids counters next to tracing domains on this pool
        return None, "this run is itself under a profiler (ROCPROF* / ROCP_* in the environment): no nested counter passes"
    work = tempfile.mkdtemp(prefix="jenga_bench_pmc_", dir="/tmp")
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child"] + argv
    env = dict(os.environ, TMPDIR="/tmp")
    per_ctr, order = {}, None
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(work, ctr)
            with open(os.path.join(work, ctr + ".out"), "w") as fo, open(os.path.join(work, ctr + ".err"), "w") as fe:
                r = subprocess.run([rp, "--pmc", ctr, "-d", d, "-o", "p", "--"] + child, cwd="/tmp", env=env, stdout=fo,
                                   stderr=fe, timeout=timeout_s)
            if r.returncode != 0:
                return None, f"the {ctr} pass exited with {r.returncode}: " + open(os.path.join(work, ctr + ".err")).read()[-300:]
            lines = [ln for ln in open(os.path.join(work, ctr + ".out")) if ln.startswith('{"pmc_child"')]
            if not lines:
                return None, f"the {ctr} pass printed no launch list"
            launches = json.loads(lines[-1])["pmc_child"]
            # (two runs of the same seeds do not keep bit-identical lists: hipBLASLt's stream-K GEMM accumulates in an order
            # that varies from run to run, a handful of borderline blocks flip -- so every counter is divided by the pairs of
            # ITS OWN pass; the passes must agree on the launches and their drop rates, and on the pairs within 2 %)
            if order is not None and ([x[0] for x in order] != [x[0] for x in launches] or any(
                    abs(x[1] - y[1]) > 0.02 * max(x[1], 1) for x, y in zip(order, launches))):
                return None, "the two passes disagree on the attention launches (count, drop rates or kept pairs beyond 2 %)"
            order = order or launches
            per_ctr[ctr + "_pairs"] = [x[1] for x in launches]
            dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
            if not dbs:
                return None, f"the {ctr} pass left no database"
            con = sqlite3.connect(dbs[0])
            tabs = [r_[0] for r_ in con.execute("select name from sqlite_master where type='table'")]
            g_ = lambda k: [t for t in tabs if k in t][0]
            q = f"""select d.id, sum(e.value) from {g_('pmc_event')} e
                    join {g_('info_pmc')} p on e.pmc_id = p.id join {g_('kernel_dispatch')} d on e.event_id = d.event_id
                    join {g_('info_kernel_symbol')} s on d.kernel_id = s.id
                    where s.kernel_name like '%bsattn_l%' and p.name = '{ctr}' group by d.id order by d.start"""
            vals = [v for _, v in con.execute(q)]
            con.close()
            if len(vals) != len(launches):
                return None, f"{ctr}: {len(vals)} attention dispatches in the database, {len(launches)} launches in the child"
            per_ctr[ctr] = vals
        out = {}
        for (tag, _), f_, fp_, w_, wp_ in zip(order, per_ctr["FETCH_SIZE"], per_ctr["FETCH_SIZE_pairs"], per_ctr["WRITE_SIZE"],
                                              per_ctr["WRITE_SIZE_pairs"]):
            t = out.setdefault(tag, dict(fetch_bytes=0.0, write_bytes=0.0, pairs=0, pairs_write_pass=0, launches=0))
            t["fetch_bytes"] += 2.0 * f_ * 1024.0
            t["write_bytes"] += w_ * 1024.0
            t["pairs"] += int(fp_)
            t["pairs_write_pass"] += int(wp_)
            t["launches"] += 1
        return out, "ok"
    except subprocess.TimeoutExpired:
        return None, f"a counter pass exceeded {timeout_s} s"
    except Exception as e:      # noqa: BLE001 - a measurement convenience must not end the run
        return None, repr(e)[:300]
    finally:
        shutil.rmtree(work, ignore_errors=True)


def _apply_pmc(read, ps, traffic_per_rate, t_pmc):
    """Counter-pass results per drop rate -> (traffic, traffic_TBps, traffic_per_rate, provenance) of the timed launches, or None
    when a drop rate of the timed launches has no counter-pass launch."""
    tot_bytes, per_rate_new = 0.0, {}
    for tag, bt in ps.get("by_tag", {}).items():
        r_ = read.get(tag)
        if r_ is None or r_["pairs"] == 0:
            continue
        bpp = r_["fetch_bytes"] / r_["pairs"] + r_["write_bytes"] / max(r_.get("pairs_write_pass", r_["pairs"]), 1)
        per_launch = bpp * bt["pairs"] / max(bt["launches"], 1)
        tot_bytes += bpp * bt["pairs"]
        per_rate_new[str(tag)] = {
            "launches": bt["launches"], "avg_launch_ms": round(bt["total_ms"] / max(bt["launches"], 1), 3),
            "kept_block_pairs_per_launch": bt["pairs"] // max(bt["launches"], 1),
            "counter_pass_launches": r_["launches"],
            "counter_pass_kept_block_pairs_per_launch": r_["pairs"] // max(r_["launches"], 1),
            "fetch_bytes_per_launch_counter_pass": int(r_["fetch_bytes"] / r_["launches"]),
            "write_bytes_per_launch_counter_pass": int(r_["write_bytes"] / r_["launches"]),
            "bytes_per_kept_pair": round(bpp),
            "bytes_per_kept_pair_from_committed_constants": traffic_per_rate.get(str(tag), {}).get("bytes_per_kept_pair"),
            "traffic_per_launch": int(per_launch),
            "traffic_TBps": round(per_launch / max(bt["total_ms"] / max(bt["launches"], 1) * 1e-3, 1e-12) / 1e12, 3),
            "ratio_to_algorithmic_bytes": round(per_launch / ATTN_ALGORITHMIC_BYTES, 1)}
    if tot_bytes <= 0 or len(per_rate_new) != len(ps.get("by_tag", {})):
        return None
    traffic = int(tot_bytes / ps["launches"])
    tbps = round(traffic / (ps["total_ms"] / ps["launches"] * 1e-3) / 1e12, 3) if ps["total_ms"] > 0 else None
    prov = ("read in this run: two child passes of this command on this box (`rocprofv3 --pmc FETCH_SIZE` and `--pmc "
            "WRITE_SIZE`, counters only) ran one computed step per stage with the same seeds -- the same kept lists, see "
            "counter_pass_kept_block_pairs_per_launch -- and read both counters for every attention launch; bytes = 2 x FETCH_SIZE "
            "KiB + WRITE_SIZE KiB (gfx950 tallies a 128-B request as 64 B: MI355X_MICROARCH.md), per kept pair and drop rate, x the "
            f"timed launches' pairs ({time.perf_counter() - t_pmc:.0f} s for both passes)")
    return traffic, tbps, per_rate_new, prov
