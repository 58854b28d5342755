"""CPU: bench.py's launcher-independent --gpus N (VERDICT r2 missing #5) and the evidence chain of roofline.traffic
(the PMC record bench.py reads must carry what it reads)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bare_gpus_n_relaunches_under_torch_distributed_run():
    import bench
    cmd = bench.launch_command(8, ["--gpus", "8", "--steps", "6", "--warmup", "1"], port=29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "6", "--warmup", "1"]
    # a free port is picked when none is given
    assert int(bench.launch_command(2, [])[bench.launch_command(2, []).index("--master-port") + 1]) > 0


def test_pmc_record_carries_the_per_pair_traffic_bench_reads():
    import bench
    pmc = json.load(open(os.path.join(ROOT, "profiles", bench.PMC_FILE)))
    assert set(pmc["rates"]) >= {"0.7", "0.8"}          # both drop rates of the Base preset, one box
    for rate in (0.7, 0.8):
        lo = bench.traffic_bytes_per_pair(pmc, rate, 0.0)
        hi = bench.traffic_bytes_per_pair(pmc, rate, 1.0)
        mid = bench.traffic_bytes_per_pair(pmc, rate, 0.71)
        # one kept pair stages a 64 KiB K/V block pair at most once: the figure must be of that order, and more overlap
        # between adjacent query blocks means more L2 hits, i.e. fewer memory-side bytes
        assert 8e3 < hi["bytes_per_pair"] <= mid["bytes_per_pair"] <= lo["bytes_per_pair"] <= 70e3
        assert mid["rate"] == rate
    assert bench.traffic_bytes_per_pair(pmc, 0.75, 0.5)["rate"] in (0.7, 0.8)
    assert abs(bench.ATTN_ALGORITHMIC_BYTES / 2.86e9 - 1) < 0.03


def test_presets_cover_the_reference_scripts_and_dense():
    import bench
    for name in ("base", "turbo", "flash", "3stage", "base-mgpu", "turbo-mgpu", "flash-mgpu", "3stage-mgpu", "dense"):
        p = bench.PRESETS[name]
        assert len(p["res"]) == len(p["steps"]) == len(p["rates"]) == len(p["shifts"])
    assert bench.PRESETS["dense"]["rates"] == [0.0, 0.0] and bench.PRESETS["dense"]["skip"] is False
    assert bench.stage_of(25, [25, 50]) == 0 and bench.stage_of(26, [25, 50]) == 1


def test_bench_line_helpers_and_keys():
    """The measurement objects the driver record must carry: the helpers exist, the FLOP
    model of a computed step matches the hand count, and the source assembles the keys."""
    import bench
    src = open(os.path.join(ROOT, "bench.py")).read()
    for key in ('"roofline_secondary"', '"loop"', '"power"', '"traffic_provenance"', '"traffic_per_rate"', '"roofline_xgmi"', '"cpu_baseline"', '"extra"',
                '"dense_reference"', '"attn_other_kernel"', '"wan14b"'):
        assert f"res[{key}]" in src or f"{key}:" in src or f"[{key}]" in src or f"setdefault({key}" in src, key
    # one computed forward at the 720p shape: 1.57 PFLOP of GEMMs
    fl = bench.hy_gemm_flops_per_computed_step(115200, 256, 20, 40)
    assert abs(fl / 1.57e15 - 1) < 0.01, fl
    ps = bench.PowerSampler(period=0.01).start()
    rec = ps.stop()
    assert "available" in rec
    assert bench.WAN_RATE_PRIORITY[:2] == [0.7, 0.8]
    # Wan2.1-14B forward: 40 layers x (6 dim^2 + 2 dim*ffn) x 2L flops
    w = bench.wan_gemm_flops_per_forward(75600, 5120, 13824, 40)
    assert 1.7e15 < w < 2.0e15, w


def test_stdout_carries_the_json_line_only_when_rccl_is_up():
    """RCCL writes a version banner to stdout through C stdio when its communicator is created; in a run that initialises it
    (N > 1, or JENGA_BENCH_FORCE_DIST=1) the banner used to land AFTER the JSON line.  bench.py points fd 1 at stderr while the
    communicator comes up, tears the process group down and flushes libc's buffers before it prints."""
    import bench
    src = open(os.path.join(ROOT, "bench.py")).read()
    i_guard, i_init = src.index("stdout_guard = _StdoutToStderr()"), src.index('dist.init_process_group("nccl"')
    i_restore = src.index("stdout_guard.restore()")
    assert i_guard < i_init < i_restore
    tail = src[src.index("# The JSON line has to be the LAST thing on stdout."):]
    assert tail.index("dist.destroy_process_group()") < tail.index("sys.stdout.write(json.dumps(res)")
    # the guard really moves fd 1 and puts it back
    r, w = os.pipe()
    saved1 = os.dup(1)
    os.dup2(w, 1)
    try:
        g = bench._StdoutToStderr()
        os.write(1, b"to-stderr")           # lands on fd 2, not in the pipe
        g.restore()
        os.write(1, b"json")
    finally:
        os.dup2(saved1, 1)
        os.close(saved1)
        os.close(w)
    assert os.read(r, 100) == b"json"
    os.close(r)


def test_in_run_counter_passes_are_parsed_and_applied(tmp_path, monkeypatch):
    """bench.py --pmc (round 6): roofline.traffic read in the run that reports it.  A stand-in `rocprofv3` (this test runs without
    a GPU) writes what the real one writes -- a rocpd sqlite database with per-dispatch counter values -- and prints the child's
    launch list; the parsing, the gfx950 corrections (2 x FETCH_SIZE KiB + WRITE_SIZE KiB), the per-rate aggregation and the
    failure paths (no tool, mismatching dispatch count) are checked here, the real passes on the GPU box."""
    import sqlite3
    import stat

    import bench
    fake = tmp_path / "rocprofv3"
    fake.write_text(f"""#!{sys.executable}
import json, os, sqlite3, sys
a = sys.argv[1:]
ctr, d = a[a.index("--pmc") + 1], a[a.index("-d") + 1]
os.makedirs(os.path.join(d, "host"), exist_ok=True)
con = sqlite3.connect(os.path.join(d, "host", "1_results.db"))
con.execute("create table rocpd_info_pmc_x (id integer, name text)")
con.execute("create table rocpd_pmc_event_x (event_id integer, pmc_id integer, value real)")
con.execute("create table rocpd_kernel_dispatch_x (id integer, event_id integer, kernel_id integer, start integer)")
con.execute("create table rocpd_info_kernel_symbol_x (id integer, kernel_name text)")
con.execute("insert into rocpd_info_pmc_x values (1, ?)", (ctr,))
con.execute("insert into rocpd_info_kernel_symbol_x values (1, 'void jenga::bsattn_lp_kernel<jenga::BF16, 4>(LpParams)')")
con.execute("insert into rocpd_info_kernel_symbol_x values (2, 'Cijk_gemm')")
n = int(os.environ.get("FAKE_LAUNCHES", "4"))
for i in range(n):                      # two XCC instances per dispatch: the query has to SUM them
    for inst in range(2):
        con.execute("insert into rocpd_pmc_event_x values (?, 1, ?)", (i, (1000.0 if ctr == "FETCH_SIZE" else 10.0) * (1 + i // 2)))
    con.execute("insert into rocpd_kernel_dispatch_x values (?, ?, 1, ?)", (i, i, 100 - i if os.environ.get("FAKE_REVERSED") else i))
con.execute("insert into rocpd_pmc_event_x values (99, 1, 5e9)")
con.execute("insert into rocpd_kernel_dispatch_x values (99, 99, 2, 50)")
con.commit()
print(json.dumps({{"pmc_child": [[0.7, 100], [0.7, 100], [0.8, 50], [0.8, 50]]}}))
""")
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", str(tmp_path) + os.pathsep + os.environ["PATH"])
    read, note = bench.pmc_read_in_this_run(["--preset", "base"], timeout_s=60)
    assert note == "ok", note
    # rate 0.7: two launches x two instances x 1000 KiB fetch -> 2 x 2 x 2000 KiB; write 2 x 2 x 10 KiB
    assert read[0.7] == dict(fetch_bytes=2.0 * 4000 * 1024, write_bytes=40.0 * 1024, pairs=200, pairs_write_pass=200, launches=2)
    assert read[0.8]["fetch_bytes"] == 2.0 * 8000 * 1024 and read[0.8]["pairs"] == 100
    ps = {"launches": 6, "total_ms": 60.0, "by_tag": {0.7: dict(launches=4, total_ms=44.0, pairs=400),
                                                      0.8: dict(launches=2, total_ms=16.0, pairs=100)}}
    import time
    traffic, tbps, per_rate, prov = bench._apply_pmc(read, ps, {"0.7": {"bytes_per_kept_pair": 1}}, time.perf_counter())
    bpp7 = (2.0 * 4000 * 1024 + 40 * 1024) / 200
    bpp8 = (2.0 * 8000 * 1024 + 80 * 1024) / 100
    assert per_rate["0.7"]["bytes_per_kept_pair"] == round(bpp7) and per_rate["0.8"]["bytes_per_kept_pair"] == round(bpp8)
    assert traffic == int((bpp7 * 400 + bpp8 * 100) / 6) and prov.startswith("read in this run")
    assert per_rate["0.7"]["bytes_per_kept_pair_from_committed_constants"] == 1
    # a drop rate of the timed launches without a counter-pass launch: not applied
    assert bench._apply_pmc({0.7: read[0.7]}, ps, {}, time.perf_counter()) is None
    # dispatch count and launch list disagree -> refused, with the reason
    monkeypatch.setenv("FAKE_LAUNCHES", "3")
    read2, note2 = bench.pmc_read_in_this_run([], timeout_s=60)
    assert read2 is None and "attention dispatches" in note2
    # no tool on the box
    monkeypatch.setenv("PATH", "/nonexistent")
    monkeypatch.setattr(bench.os.path, "exists", lambda p: False if "rocprofv3" in p else os.path.lexists(p))
    read3, note3 = bench.pmc_read_in_this_run([], timeout_s=5)
    assert read3 is None and "not found" in note3


def test_bench_and_its_helper_modules_reference_no_undefined_global():
    """bench.py was split into benchlib/ in round 6; almost none of it runs without a GPU, so a name lost in the move would
    only show on the GPU box.  tools/check_names.py: every name a function resolves as a module global exists."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_names
    for m in ("bench", "benchlib.consts", "benchlib.pmc", "benchlib.launch", "benchlib.power", "benchlib.secondary",
              "benchlib.cpu_ref", "benchlib.wan"):
        assert check_names.undefined_globals(m) == [], m
