#!/usr/bin/env python3
"""Benchmark of the north-star metric: HunyuanVideo DiT denoising loop, sec/video (720x1280, 125 frames, 50 steps)
at Jenga-Base settings, on N GPUs of one node (Ulysses sequence parallelism over RCCL for N > 1).

    python bench.py --gpus 1 --steps 6 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...        (no launcher: re-executes itself under torch.distributed.run, same thing)

A "step" is one scheduler step of the 50-step loop = one call of the (synthetic-weight, full-size) DiT with the
Jenga forward: Hilbert gather -> 20 double + 40 single blocks (QKV GEMMs, fused RMSNorm+RoPE, block selection,
block-sparse attention, proj/MLP GEMMs) -> scatter, or, on the 27 steps outside jenga_hyvideo.py:28's
non_skip_steps, the cached-residual shortcut.  With --steps 50 the real schedule is run; with fewer steps a
class-balanced sample of it is timed (computed@rate0, skipped, computed@rate1) and sec/video is
12*t(computed@rate0) + 11*t(computed@rate1) + 27*t(skipped) -- the JSON says which.

One JSON line on rank 0, with `roofline` (block-sparse attention kernel: algorithmic FLOPs of the realised masks /
launch durations measured with HIP events on the launch stream) and `cpu_baseline` (the oracle's AttenCarve op timed
on the host cores on a bounded sample and extrapolated by kept block pairs; attention only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


PRESETS = {   # scripts/hyvideo_jenga_{base,turbo,flash,3stage}.sh and scripts/hyvideo_multigpu_jenga_*.sh
    # base: BASELINE.json configs[1] quotes Jenga-Base at sa-drop 0.7/0.8 (the reference's README table); the shipped
    # single-GPU script uses 0.75/0.85 (less attention work) = "base-mgpu" here (the 8-GPU script has the same rates)
    "base": dict(res=[1.0, 1.0], steps=[0.5, 1.0], rates=[0.7, 0.8], shifts=[7, 7], p=0.3),
    "turbo": dict(res=[0.75, 1.0], steps=[0.5, 1.0], rates=[0.7, 0.8], shifts=[7, 9], p=0.3),
    "flash": dict(res=[0.75, 1.0], steps=[0.5, 1.0], rates=[0.8, 0.95], shifts=[7, 9], p=0.5),
    "3stage": dict(res=[0.5, 0.75, 1.0], steps=[0.3, 0.5, 1.0], rates=[0.75, 0.85, 0.85], shifts=[7, 9, 11], p=0.3),
    "base-mgpu": dict(res=[1.0, 1.0], steps=[0.5, 1.0], rates=[0.75, 0.85], shifts=[7, 7], p=0.3),
    "turbo-mgpu": dict(res=[0.75, 1.0], steps=[0.5, 1.0], rates=[0.75, 0.85], shifts=[7, 9], p=0.3),
    "flash-mgpu": dict(res=[0.75, 1.0], steps=[0.5, 1.0], rates=[0.8, 0.95], shifts=[7, 9], p=0.5),
    "3stage-mgpu": dict(res=[0.5, 0.75, 1.0], steps=[0.3, 0.5, 1.0], rates=[0.75, 0.85, 0.85], shifts=[7, 9, 11], p=0.3),
    # the un-accelerated model (README.md:80-82 "HunyuanVideo" row, 1625 s on one H800): sa-drop 0 takes the dense
    # branch of the blocks (models_mul...:254-258), no step skipping (enable_skip is a Jenga switch): 50 computed steps
    "dense": dict(res=[1.0, 1.0], steps=[0.5, 1.0], rates=[0.0, 0.0], shifts=[7, 7], p=0.3, skip=False),
}

from benchlib.consts import ATTN_ALGORITHMIC_BYTES, FLOPS_PER_PAIR, HBM_PEAK_GBPS, MFMA_PEAK_TFLOPS, PMC_FILE, attn_kernel_name  # noqa: E402,F401
from benchlib.cpu_ref import cpu_baseline  # noqa: E402
from benchlib.launch import _flush_c_stdio, _free_port, _StdoutToStderr, launch_command  # noqa: E402,F401
from benchlib.pmc import _apply_pmc, pmc_read_in_this_run, traffic_bytes_per_pair  # noqa: E402
from benchlib.power import PowerSampler, _host_cpu  # noqa: E402,F401
from benchlib.secondary import _timed, hy_gemm_flops_per_computed_step, secondary_roofline  # noqa: E402,F401
from benchlib.wan import WAN_RATE_PRIORITY, wan_extra, wan_gemm_flops_per_forward, wan_main, wan_setup  # noqa: E402,F401


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rates", type=float, nargs="+", default=None,
                    help="sa-drop-rates per stage; default per preset (base: 0.7 0.8 = BASELINE.json configs[1]; "
                         "0.75 0.85 = the shipped scripts/hyvideo_jenga_base.sh)")
    ap.add_argument("--preset", choices=sorted(PRESETS), default="base",
                    help="scripts/hyvideo[_multigpu]_jenga_{base,turbo,flash,3stage}.sh: resolution / step / shift / "
                         "drop-rate / p-remain lists (the *-mgpu presets carry the rate sets of the 8-GPU scripts)")
    ap.add_argument("--p-remain", type=float, default=None, help="default: the preset's")
    ap.add_argument("--peaky", type=float, default=0.0, metavar="GAIN",
                    help="> 0: scale the Q/K RMSNorm weights by GAIN (block scores x GAIN^2).  Random weights give "
                         "nearly flat pooled scores, where the p-remain rule keeps ~p of all blocks; a trained model's "
                         "block softmax is peaked and top_k decides (SURVEY.md 8(d) asks for both regimes). "
                         "The realised kept block pairs per launch are in the JSON either way")
    ap.add_argument("--latent", type=int, nargs=3, default=[32, 90, 160], help="latent T H W (720x1280x125f)")
    ap.add_argument("--depth", type=int, nargs=2, default=None, help="override (double, single) block counts (debug only)")
    ap.add_argument("--valid-text", type=int, default=64)
    ap.add_argument("--i2v", action="store_true",
                    help="HunyuanVideo-I2V flavour: token_replace modulation of the first latent frame, 512 text tokens "
                         "(4 text blocks) -- BASELINE.json configs[4] with --preset 3stage")
    ap.add_argument("--gemm-tuning", default="auto", metavar="FILE|auto|off|record:FILE",
                    help="hipBLASLt solution selection for the dense GEMMs (jenga_amd/gemm_tuning.py): auto = replay "
                         "jenga_amd/tuned/hipblaslt_gfx950.csv when present (no timing at run time), off = the library's "
                         "default heuristic, record:FILE = let TunableOp time every solution for the shapes this run "
                         "meets and write FILE (slow; not a measurement run)")
    ap.add_argument("--workload", choices=["hy720p", "wan14b"], default="hy720p",
                    help="hy720p = BASELINE.json configs[1] (the headline); wan14b = configs[3]: Wan2.1-14B T2V 1280x720x81f "
                         "Jenga-Base (scripts/wan_14B_jenga_base.sh), a step = one scheduler step = two CFG forwards")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip roofline_secondary (isolated HIP-event timings of the bandwidth-bound kernels and the GEMM "
                         "classes at the workload's shapes, run after the timed region)")
    ap.add_argument("--no-wan-extra", action="store_true",
                    help="skip the short Wan2.1-14B leg (one forward at each drop rate, after the timed region) that puts a "
                         "configs[3] number into the default N=1 record")
    ap.add_argument("--no-xgmi-extras", action="store_true",
                    help="N > 1: skip the legs behind the timed region that fill `roofline_xgmi` (the exchanges timed alone, one "
                         "computed step per stage without the fabric, rank 0's single-rank steps for `efficiency`)")
    ap.add_argument("--no-other-kernel-ref", action="store_true",
                    help="skip the computed steps re-run after the timed region with the OTHER attention kernel (LP <-> pair)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pmc", choices=["auto", "on", "off"], default="auto",
                    help="roofline.traffic READ IN THIS RUN: after the timed region, two child passes of this script under "
                         "`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (counters only, no tracing domain) run one computed "
                         "step per stage with the same seeds and read the memory-side bytes of every attention launch.  auto = on "
                         "for a single-GPU hy720p run when rocprofv3 is on the box; a failed pass falls back to the committed "
                         "per-pair constants and says so")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-dense-ref", action="store_true",
                    help="skip the ONE dense (sa-drop 0) computed step that is run after the timed region to report "
                         "speedup_vs_dense (the reference's own headline: 1625 s -> 310 s, README.md:80-82)")
    ap.add_argument("--coherent", type=float, default=0.0, metavar="SMOOTH",
                    help="> 0: spatially smooth synthetic latents (white noise at 1/SMOOTH of the latent resolution, "
                         "upsampled trilinearly, + 10 %% white noise) instead of iid noise: Hilbert-adjacent query blocks "
                         "then see similar keys and keep similar block lists, as a trained model's do (SURVEY.md 8(d))")
    ap.add_argument("--sim-exchange-gbps", type=float, default=0.0, metavar="G",
                    help="with --simulate-ranks: every exchange takes latency + (bytes leaving the rank) / G GB/s on a side "
                         "stream (0 = plain local copies on the compute stream).  300 is what an 8-GPU xGMI all-to-all is "
                         "assumed to sustain per GPU (7 links x ~77 GB/s per direction = 537 GB/s peak)")
    ap.add_argument("--sim-exchange-latency-us", type=float, default=15.0,
                    help="fixed cost per simulated collective (launch + rendezvous)")
    ap.add_argument("--simulate-ranks", type=int, default=0,
                    help="diagnostic, single process: run rank 0's share of an N-rank Ulysses job (per-rank shapes, "
                         "pack / unpack kernels, 24/N heads) with the exchanges replaced by local copies -- the "
                         "compute-side time of one rank; NOT a multi-GPU measurement")
    return ap.parse_args()


def stage_of(i, split):
    """Step i runs at stage k = number of split points strictly below i (the switch happens AFTER step split[k],
    pipeline_hunyuan_video_prores.py:697-698)."""
    k = 0
    while k < len(split) - 1 and i > split[k]:
        k += 1
    return k


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # called bare (`python bench.py --gpus 8 ...`): become the launcher -- one rank per GPU under
        # torch.distributed.run, rendezvous on 127.0.0.1 (the reference's scripts/hyvideo_multigpu_jenga_base.sh:7 uses
        # torchrun the same way).  Under an external torchrun WORLD_SIZE is set and this branch is not taken.
        cmd = launch_command(a.gpus, sys.argv[1:])
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.stdout.flush()
        os.execv(cmd[0], cmd)
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch one rank per GPU")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    # JENGA_BENCH_FORCE_DIST=1: take every multi-rank code path with a world of ONE rank on RCCL (process group, exchange
    # warm-up, sequence-parallel modules, choice broadcast, max-over-ranks reductions) -- the smoke test of the N > 1 launch
    # on a one-GPU box; the numbers of such a run are a single-GPU measurement with the sequence-parallel call structure
    dist_on = world > 1 or os.environ.get("JENGA_BENCH_FORCE_DIST", "0") == "1"
    stdout_guard = None
    if dist_on:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
        # RCCL prints a five-line version banner to STDOUT (C stdio) when its communicator comes up: until the communicator
        # exists and libc's buffers are flushed, file descriptor 1 points at stderr, so that stdout carries the JSON line only
        stdout_guard = _StdoutToStderr()
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    sim = a.simulate_ranks if not dist_on else 0
    if sim > 1:
        from torch.testing._internal.distributed.fake_pg import FakeStore
        dist.init_process_group(backend="fake", rank=0, world_size=sim, store=FakeStore())

        def _gather_into(out, x, group=None, **kw):          # rank 0's view: every peer holds what I hold
            out.view((sim,) + tuple(x.shape)).copy_(x.unsqueeze(0).expand((sim,) + tuple(x.shape)))
        dist.all_gather_into_tensor = _gather_into

    class _EventWait:
        def __init__(self, ev):
            self.ev = ev

        def wait(self):
            torch.cuda.current_stream().wait_event(self.ev)
            return True

    class _LocalExchange:
        """--simulate-ranks: the exchanges of rank 0 replaced by local copies of the same shapes.  With
        --sim-exchange-gbps G the copy runs on a SIDE stream behind a delay of latency + (bytes that would leave this
        rank) / G (jenga_stream_delay: one wavefront), and .wait() makes the compute stream wait for it -- the same
        dependency structure as an RCCL exchange on the process group's stream, so what the blocks enqueue between
        posting an exchange and waiting for it overlaps the simulated transfer exactly as it would overlap a real one."""

        def __init__(self, n, gbps=0.0, latency_us=0.0):
            self.n, self.gbps, self.lat = n, gbps, latency_us
            self.side = torch.cuda.Stream() if gbps > 0 else None
            self.sim_us = 0.0          # simulated transfer time posted so far (host-side sum)

        def size(self):
            return self.n

        def rank(self):
            return 0

        def _post(self, copy, nbytes_out):
            if self.side is None:
                copy()
                return _Done()
            us = self.lat + nbytes_out / (self.gbps * 1e3)
            self.sim_us += us
            ready = torch.cuda.Event()
            ready.record()
            self.side.wait_event(ready)
            with torch.cuda.stream(self.side):
                _capi.stream_delay(us, stream=self.side)
                copy()
                done = torch.cuda.Event()
                done.record(self.side)
            return _EventWait(done)

        def all_to_all(self, recvs, sends):
            def copy():
                for rc, sd in zip(recvs, sends):
                    rc.copy_(sd)
            out = sum(sd.numel() * sd.element_size() for sd in sends) * (self.n - 1) // self.n
            return [self._post(copy, out)]

        def all_gather(self, out, x):
            return self._post(lambda: out.copy_(x.unsqueeze(0).expand_as(out)),
                              x.numel() * x.element_size() * (self.n - 1))

    class _Done:
        def wait(self):
            return True

    from jenga_amd import _capi, gemm_tuning
    from jenga_amd.dit import NON_SKIP_STEPS, JengaHYVideoDiT
    if a.workload == "wan14b":
        if dist_on or sim > 1:
            raise SystemExit("--workload wan14b is a single-GPU line (the reference's multi-GPU Wan path is USP / FSDP, not "
                             "Jenga-aware: SURVEY.md §2)")
        return wan_main(a, dev)
    gemm_file = None
    if a.gemm_tuning.startswith("record:"):
        gemm_file = gemm_tuning.enable(a.gemm_tuning[len("record:"):], tune=True)
    elif a.gemm_tuning != "off":
        gemm_file = gemm_tuning.enable(None if a.gemm_tuning == "auto" else a.gemm_tuning)
    from jenga_amd.modules import ulysses

    torch.manual_seed(0)
    kw = {}
    if a.depth:
        kw = dict(depth_double=a.depth[0], depth_single=a.depth[1])
    model = JengaHYVideoDiT(dtype=torch.bfloat16, device=dev, **kw).init_synthetic_weights(0.02, seed=0)
    if dist_on:
        # create the RCCL communicator and its channels now, whatever --warmup says (lazy creation costs seconds)
        w_ = torch.ones(world, 16, device=dev)
        r_ = torch.empty_like(w_)
        dist.all_to_all_single(r_, w_)
        dist.all_gather_into_tensor(torch.empty(world * 16, device=dev), w_[0].contiguous())
        ops = []
        for step in range(1, world):     # the grouped send/recv the Ulysses exchange uses: create its channels now too
            ops.append(dist.P2POp(dist.isend, w_[(rank + step) % world], (rank + step) % world))
            ops.append(dist.P2POp(dist.irecv, r_[(rank - step) % world], (rank - step) % world))
        for w__ in (dist.batch_isend_irecv(ops) if ops else []):      # (no peers in a world of one rank)
            w__.wait()
        dist.all_reduce(w_)
        torch.cuda.synchronize()
    if stdout_guard is not None:
        stdout_guard.restore()
    sim_ex = _LocalExchange(sim, a.sim_exchange_gbps, a.sim_exchange_latency_us) if sim > 1 else None
    if dist_on or sim > 1:
        ulysses.init_sequence_parallel()
        for blk in list(model.double_blocks) + list(model.single_blocks):
            blk.hybrid_seq_parallel_attn = ulysses.UlyssesAttenCarve(exchange=sim_ex)
    from jenga_amd import prores
    preset = dict(PRESETS[a.preset])
    if a.rates:
        preset["rates"] = list(a.rates) + [a.rates[-1]] * (len(preset["res"]) - len(a.rates))
    a.rates = preset["rates"]
    if a.p_remain is None:
        a.p_remain = preset["p"]
    if a.peaky > 0:
        with torch.no_grad():
            for name, p_ in model.named_parameters():
                if name.endswith(("q_norm.weight", "k_norm.weight")):
                    p_.mul_(a.peaky)
    T, Hh, W = a.latent
    shapes, split = prores.stage_plan((T, Hh, W), 50, preset["res"], preset["steps"])
    g = torch.Generator(device=dev).manual_seed(42)
    stages = []
    amps = prores.stage_text_amps(shapes)   # stage 0 carries the amplifier, every later stage 0.0 (:577, :755)

    def synth_latents(shp):
        if a.coherent > 0:
            lo = [max(2, int(round(d / a.coherent))) for d in shp]
            base_ = torch.randn(1, 16, *lo, generator=g, device=dev, dtype=torch.float32)
            x_ = torch.nn.functional.interpolate(base_, size=list(shp), mode="trilinear", align_corners=False)
            x_ = x_ / x_.std() + 0.1 * torch.randn(1, 16, *shp, generator=g, device=dev, dtype=torch.float32)
            return x_.to(torch.bfloat16)
        return torch.randn(1, 16, *shp, generator=g, device=dev, dtype=torch.bfloat16)

    for k, shp in enumerate(shapes):      # static geometry + synthetic latents per resolution stage
        cos_k, sin_k = model.set_stage(shp, dev)
        stages.append(dict(shape=shp, cos=cos_k, sin=sin_k, curve=model.curve_sel, l2h=model.linear_to_hilbert,
                           h2l=model.hilbert_order,
                           latents=synth_latents(shp),
                           text_amp=amps[k]))
    # steps forced to compute by `start_stage` right after a resolution switch (:755)
    forced = {split[k] + 1 for k in range(len(split) - 1) if preset["res"][k] != 1.0}
    do_skip = preset.get("skip", True)
    computed_steps = sorted(set(NON_SKIP_STEPS) | forced) if do_skip else list(range(50))
    sched = prores.FlowMatchSchedule(50, shift=preset["shifts"][0])
    g2 = torch.Generator(device=dev).manual_seed(43)
    n_txt = 512 if a.i2v else 256
    text = torch.randn(1, n_txt, 4096, generator=g2, device=dev, dtype=torch.bfloat16)
    text2 = torch.randn(1, 768, generator=g2, device=dev, dtype=torch.bfloat16)
    text_mask = torch.zeros(1, n_txt, dtype=torch.int64, device=dev)
    text_mask[:, : a.valid_text] = 1
    guidance = torch.tensor([6000.0], device=dev)
    model.p_remain_rates = a.p_remain
    model.i2v_condition_type = "token_replace" if a.i2v else None
    model.text_amp = 0.0
    model.num_steps = 50
    model.enable_skip = do_skip

    def run_step(i):
        k = stage_of(i, split)
        st = stages[k]
        model.curve_sel, model.linear_to_hilbert, model.hilbert_order = st["curve"], st["l2h"], st["h2l"]
        model.cnt = i
        model.sa_drop_rate = a.rates[k]
        if _capi.ATTN_PROFILE is not None:
            _capi.ATTN_PROFILE.tag = a.rates[k]
        model.text_amp = st["text_amp"]
        model.start_stage = i in forced
        tval = sched.timesteps[i:i + 1].to(dev)
        out_ = model(st["latents"], tval, text_states=text, text_mask=text_mask, text_states_2=text2,
                     freqs_cos=st["cos"], freqs_sin=st["sin"], guidance=guidance, return_dict=False)
        if i in split[:-1] and k + 1 < len(stages) and preset["res"][k] != 1.0:
            # the re-noising hop to the next resolution (x0-predict, trilinear upsample, add noise)
            nxt = stages[k + 1]
            prores.switch_stage(sched, out_, i, st["latents"], nxt["shape"], preset["shifts"][k + 1], nxt["latents"])
        return out_

    def klass(i):
        return (stage_of(i, split), "c" if i in computed_steps else "s")

    if a.steps >= 50:
        plan = [i % 50 for i in range(a.steps)]      # whole loop(s); sec/video = elapsed * 50 / steps
        sampled = False
    else:
        # class-balanced sample of the schedule: one computed + one skipped step of EVERY stage first, then a second
        # computed step per stage -- so that the default K = 6 covers every class of a three-stage preset
        comp_k, skip_k = [], []
        for k in range(len(stages)):
            ids = [i for i in range(50) if stage_of(i, split) == k]
            comp_k.append([i for i in ids if i in computed_steps])
            skip_k.append([i for i in ids if i not in computed_steps])
        # (a skipped step replays the residual of the last computed step, so it has to follow one of ITS stage)
        pattern = [i for c, s_ in zip(comp_k, skip_k) for i in c[:1] + s_[:1]] + [c[1] for c in comp_k if len(c) > 1]
        plan = [pattern[j % len(pattern)] for j in range(a.steps)]
        sampled = True

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    if a.pmc_child:
        # (under rocprofv3 --pmc, started by pmc_read_in_this_run of the parent run): one computed step per stage, nothing else;
        # the attention launches in dispatch order with their drop rate and kept pairs go to stdout
        _capi.ATTN_PROFILE = prof_c = _capi.AttnProfile()
        for k in range(len(stages)):
            run_step(next(i for i in computed_steps if stage_of(i, split) == k and i not in forced))
        torch.cuda.synchronize()
        print(json.dumps({"pmc_child": [[tag, int(pr.item()) if torch.is_tensor(pr) else int(pr)] for tag, pr in prof_c.per_launch]}),
              flush=True)
        return

    if dist_on or sim > 1:
        # per-rank GEMM shapes (M = S_img / N): hipBLASLt's first pick is not its fastest there (profiles/
        # r03_gemm_tunableop.json, r03_gemm_epilogue_ab.json) -- let jenga_linear time its first 32 candidates once per
        # shape, during the untimed priming steps below (32 against 16: -0.6 % per rank-step, profiles/r04_gemm_candidates_ab.json;
        # on one GPU the first pick is the fastest: 77.16 vs 77.33 s/video with 16 timed)
        os.environ.setdefault("JENGA_GEMM_CANDIDATES", "32")
    if int(os.environ.get("JENGA_GEMM_CANDIDATES", "1")) > 1:
        for k in range(len(stages)):      # one untimed computed step per stage: every GEMM shape gets its plan here
            run_step(next(i for i in computed_steps if stage_of(i, split) == k))
        if dist_on:
            # every rank timed candidates on its own: adopt rank 0's choices everywhere, so that the replicated text stream
            # sees the same arithmetic on every rank (the imported index replaces each rank's plan; no further timing)
            # (every rank ALWAYS enters both collectives: a failed export travels as None, a failed import is counted)
            rec = None
            if rank == 0:
                try:
                    rec = _capi.linear_export_choices()
                except Exception as e:      # noqa: BLE001 - a measurement convenience must not end the run
                    print(f"bench.py: exporting rank 0's GEMM choices failed ({e!r})", file=sys.stderr)
            box = [rec]
            dist.broadcast_object_list(box, src=0)
            ok_ = 0
            if box[0] is not None:
                try:
                    _capi.linear_import_choices(box[0])
                    ok_ = 1
                except Exception as e:      # noqa: BLE001
                    print(f"bench.py: rank {rank}: adopting rank 0's GEMM choices failed ({e!r})", file=sys.stderr)
            okt = torch.tensor([ok_], device=dev, dtype=torch.int32)
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
            gemm_choices_synced = bool(okt.item())
    if "gemm_choices_synced" not in locals():
        gemm_choices_synced = None       # (single rank, or no candidate timing: nothing to synchronise)
    for w in range(a.warmup):
        run_step(computed_steps[0] if w % 2 == 0 else computed_steps[-1])   # computed steps: fills previous_residual
    barrier()
    _capi.ATTN_PROFILE = prof = _capi.AttnProfile()
    if sim_ex is not None:
        sim_ex.sim_us = 0.0
    evs = []
    pair_marks = []      # cumulative kept block pairs after every timed step (device scalars: no synchronisation)
    power = PowerSampler().start() if rank == 0 else None
    t0 = time.perf_counter()
    for i in plan:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = run_step(i)
        e1.record()
        evs.append((i, e0, e1))
        pair_marks.append(prof.pairs.clone() if prof.pairs is not None else None)
    barrier()
    elapsed = time.perf_counter() - t0
    power_rec = power.stop() if power is not None else None
    _capi.ATTN_PROFILE = None
    if sim_ex is not None:
        sim_ex.sim_us_timed = sim_ex.sim_us
    if dist_on:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    finite = bool(torch.isfinite(out.float()).all().item())

    extras_failed = {}      # legs behind the timed region that raised: reported in the line, never fatal
    # ---- the un-accelerated model beside it (NOT in the timed region): ONE computed step with sa-drop 0 at the final
    #      resolution, no skipping; its loop is 50 such steps (README.md:80-82: 1625 s dense vs 310 s Jenga-Base)
    try:
        dense_ms = None
        if not a.no_dense_ref and a.preset != "dense":
            from jenga_amd.modules import attention as _att
            st = stages[-1]
            if not dist_on and sim <= 1:
                _att._dense_lists(dev, 1, model.heads_num, (st["h2l"].numel() + n_txt) // 128)   # built once, outside the timing
            model.curve_sel, model.linear_to_hilbert, model.hilbert_order = st["curve"], st["l2h"], st["h2l"]
            model.cnt, model.sa_drop_rate, model.text_amp, model.start_stage, model.enable_skip = 0, 0.0, 0.0, False, False
            barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            model(st["latents"], sched.timesteps[0:1].to(dev), text_states=text, text_mask=text_mask, text_states_2=text2,
                  freqs_cos=st["cos"], freqs_sin=st["sin"], guidance=guidance, return_dict=False)
            e1.record()
            barrier()
            dense_ms = e0.elapsed_time(e1)
            if dist_on:
                tt = torch.tensor([dense_ms], device=dev, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dense_ms = float(tt.item())
            model.enable_skip = do_skip
    except Exception as e:      # noqa: BLE001 - an extra leg behind the timed region must not cost the line
        extras_failed['dense_reference'] = repr(e)[:300]
        print(f"bench.py: the dense_reference leg failed ({e!r}); the line goes out without it", file=sys.stderr)
        dense_ms = None
        model.enable_skip = do_skip

    # ---- the other attention kernel beside it (NOT in the timed region, not `value`): one computed step per stage with the
    #      kernel that is not the default (pair kernel <-> LP kernel), same box, same lists
    try:
        rot_ms, rot_prof, other_flags = {}, None, None
        if not a.no_other_kernel_ref and not dist_on and sim <= 1 and a.preset != "dense" and \
                (_capi.ATTN_DEFAULT_FLAGS & (_capi.ATTN_LP | _capi.ATTN_PAIR)):
            flags0 = _capi.ATTN_DEFAULT_FLAGS
            other_flags = _capi.ATTN_LP_FLAGS if (flags0 & _capi.ATTN_PAIR) else _capi.ATTN_PAIR_FLAGS
            _capi.ATTN_DEFAULT_FLAGS = other_flags
            _capi.ATTN_PROFILE = rot_prof = _capi.AttnProfile()
            try:
                for k in range(len(stages)):
                    i_k = next(i for i in computed_steps if stage_of(i, split) == k and i not in forced)
                    barrier()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    run_step(i_k)
                    e1.record()
                    barrier()
                    rot_ms[k] = e0.elapsed_time(e1)
            finally:
                _capi.ATTN_DEFAULT_FLAGS = flags0
                _capi.ATTN_PROFILE = None
    except Exception as e:      # noqa: BLE001 - an extra leg behind the timed region must not cost the line
        extras_failed['attn_other_kernel'] = repr(e)[:300]
        print(f"bench.py: the attn_other_kernel leg failed ({e!r}); the line goes out without it", file=sys.stderr)
        rot_ms, rot_prof, other_flags = {}, None, None

    # ---- N > 1: what the exchange costs, measured (NOT in the timed region).  Reference: xdit_ring_atten.py:118-131,
    #      212-217 (6 all-to-alls per layer); here per layer: Q|K, V, O all-to-alls + the text all-gather (ulysses.py)
    try:
        xgmi_raw = None
        if dist_on and not a.no_xgmi_extras and a.preset != "dense":
            sp_blocks = [b for b in list(model.double_blocks) + list(model.single_blocks) if b.hybrid_seq_parallel_attn]
            ex0 = sp_blocks[0].hybrid_seq_parallel_attn.exchange()
            Hh_, D_ = model.heads_num, 128
            Hn_ = Hh_ // world
            xgmi_raw = {"mode": ex0.mode, "stages": [], "bytes_counted": sum(b.hybrid_seq_parallel_attn.exchange().bytes_out for b in sp_blocks),
                        "exchange_calls_counted": sum(b.hybrid_seq_parallel_attn.exchange().calls for b in sp_blocks)}
            for k in range(len(stages)):
                S_img_k = stages[k]["h2l"].numel()
                S_loc_k = S_img_k // world
                mk_ = lambda *shape: torch.empty(shape, dtype=torch.bfloat16, device=dev)
                snd = [mk_(world, S_loc_k, Hn_, D_) for _ in range(4)]
                rcv = [mk_(world, S_loc_k, Hn_, D_) for _ in range(4)]
                tx, tall = mk_(1, n_txt, Hn_, D_), mk_(world, 1, n_txt, Hn_, D_)
                bytes_layer = 4 * snd[0].numel() * 2 * (world - 1) // world + tx.numel() * 2 * (world - 1)

                def one_layer():
                    ws = ex0.all_to_all(rcv[:2], snd[:2]) + ex0.all_to_all(rcv[2:3], snd[2:3])
                    for w_ in ws:
                        w_.wait()
                    ws = ex0.all_to_all(rcv[3:4], snd[3:4])
                    wt = ex0.all_gather(tall, tx)
                    for w_ in ws:
                        w_.wait()
                    wt.wait()
                for _ in range(3):
                    one_layer()
                barrier()
                reps = 20
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    one_layer()
                e1.record()
                barrier()
                alone_ms = e0.elapsed_time(e1) / reps
                # one computed step of this stage with the real exchange and one with every transfer replaced by a local copy
                i_k = next(i for i in computed_steps if stage_of(i, split) == k and i not in forced)

                def timed_step():
                    barrier()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    run_step(i_k)
                    e1.record()
                    barrier()
                    tt_ = torch.tensor([e0.elapsed_time(e1)], device=dev, dtype=torch.float64)
                    dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
                    return float(tt_.item())
                with_ms = timed_step()
                saved = [b.hybrid_seq_parallel_attn._exchange for b in sp_blocks]
                nofab = ulysses.NoFabricExchange(world, rank)
                for b in sp_blocks:
                    b.hybrid_seq_parallel_attn._exchange = nofab
                try:
                    without_ms = timed_step()
                finally:
                    for b, e_ in zip(sp_blocks, saved):
                        b.hybrid_seq_parallel_attn._exchange = e_
                xgmi_raw["stages"].append(dict(stage=k, S_loc=S_loc_k, bytes_out_per_layer=bytes_layer, exchange_alone_ms=alone_ms,
                                               step_ms=with_ms, step_ms_without_fabric=without_ms))
            # rank 0 alone: the single-rank model on the same box, one computed + one skipped step per stage (`efficiency`)
            n1 = {}
            if rank == 0:
                saved_sp = [b.hybrid_seq_parallel_attn for b in sp_blocks]
                for b in sp_blocks:
                    b.hybrid_seq_parallel_attn = None
                try:
                    for k in range(len(stages)):
                        i_c = next(i for i in computed_steps if stage_of(i, split) == k and i not in forced)
                        i_s = next((i for i in range(50) if stage_of(i, split) == k and i not in computed_steps), None)
                        run_step(i_c)                      # plans for the full-M GEMM shapes, the residual cache
                        torch.cuda.synchronize()
                        for key_, i_ in (("c", i_c), ("s", i_s)):
                            if i_ is None:
                                continue
                            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            e0.record()
                            run_step(i_)
                            e1.record()
                            torch.cuda.synchronize()
                            n1[(k, key_)] = e0.elapsed_time(e1)
                finally:
                    for b, sp_ in zip(sp_blocks, saved_sp):
                        b.hybrid_seq_parallel_attn = sp_
            barrier()
            xgmi_raw["n1"] = n1
    except Exception as e:      # noqa: BLE001 - an extra leg behind the timed region must not cost the line
        extras_failed['roofline_xgmi'] = repr(e)[:300]
        print(f"bench.py: the roofline_xgmi leg failed ({e!r}); the line goes out without it", file=sys.stderr)
        xgmi_raw = None

    cls = {}
    for i, e0, e1 in evs:
        cls.setdefault(klass(i), []).append(e0.elapsed_time(e1))
    mean = lambda v: sum(v) / len(v) if v else float("nan")
    counts = {}
    for i in range(50):
        counts[klass(i)] = counts.get(klass(i), 0) + 1
    if sampled:
        if dist_on:   # scale the per-class event times so that they sum to the max-over-ranks wall time
            scale = elapsed * 1e3 / max(sum(e0.elapsed_time(e1) for _, e0, e1 in evs), 1e-9)
        else:
            scale = 1.0
        # a class that was not sampled (very small --steps) borrows the mean of the same kind from another stage
        def class_ms(key):
            if key in cls:
                return mean(cls[key])
            same = [mean(v) for kk, v in cls.items() if kk[1] == key[1]]
            return same[-1] if same else 0.0
        sec_per_video = sum(n * class_ms(key) for key, n in counts.items()) * scale / 1e3
        unsampled = sorted(f"{k}{c}" for (k, c) in counts if (k, c) not in cls)
    else:
        sec_per_video = elapsed * 50.0 / len(plan)
        unsampled = []
    ps = prof.summary()
    # HBM-side bytes per launch: the per-kept-pair figure of the committed PMC passes (profiles/, separate rocprofv3
    # --pmc runs of this kernel on this workload; recipe tools/pmc_attn2.sh) x this run's kept pairs per launch
    traffic = traffic_tbps = None
    traffic_per_rate = {}
    pmc_rel = os.path.join("profiles", PMC_FILE)
    try:
        pmc = json.load(open(os.path.join(ROOT, pmc_rel)))
    except FileNotFoundError:
        pmc = None
        print(f"bench.py: {pmc_rel} not found: roofline.traffic stays null", file=sys.stderr)
    lq_kernel = bool(_capi.ATTN_DEFAULT_FLAGS & _capi.ATTN_PAIR)
    if pmc is not None and ps["launches"] > 0 and not lq_kernel:
        shared_now = ps.get("adjacent_shared_frac")
        tot_bytes = 0.0
        for tag, bt in ps.get("by_tag", {}).items():
            pr = traffic_bytes_per_pair(pmc, tag, shared_now)
            if pr is None:
                continue
            b_ = pr["bytes_per_pair"] * bt["pairs"]
            tot_bytes += b_
            per_launch = b_ / max(bt["launches"], 1)
            traffic_per_rate[str(tag)] = {
                "launches": bt["launches"], "avg_launch_ms": round(bt["total_ms"] / max(bt["launches"], 1), 3),
                "kept_block_pairs_per_launch": bt["pairs"] // max(bt["launches"], 1),
                "bytes_per_kept_pair": round(pr["bytes_per_pair"]), "pmc_points": pr["points"], "pmc_rate_used": pr["rate"],
                "traffic_per_launch": int(per_launch),
                "traffic_TBps": round(per_launch / max(bt["total_ms"] / max(bt["launches"], 1) * 1e-3, 1e-12) / 1e12, 3),
                "ratio_to_algorithmic_bytes": round(per_launch / ATTN_ALGORITHMIC_BYTES, 1)}
        if tot_bytes > 0:
            traffic = int(tot_bytes / ps["launches"])
            if ps["total_ms"] > 0:
                traffic_tbps = round(traffic / (ps["total_ms"] / ps["launches"] * 1e-3) / 1e12, 3)
    # ---- ... and READ in this run (review r5 item 6): child passes under rocprofv3 --pmc, same seeds and lists, this box
    traffic_provenance = ("derived, not read in this run: committed per-kept-pair constants (one per drop "
                          "rate) x this run's pairs, launch by launch")
    pmc_note = None
    want_pmc = a.pmc == "on" or (a.pmc == "auto" and not dist_on and sim <= 1 and a.preset != "dense" and rank == 0)
    if want_pmc and ps["launches"] > 0:
        child_argv = ["--preset", a.preset, "--rates"] + [str(r_) for r_ in a.rates] + [
            "--p-remain", str(a.p_remain), "--latent"] + [str(v_) for v_ in a.latent] + [
            "--valid-text", str(a.valid_text), "--peaky", str(a.peaky), "--coherent", str(a.coherent), "--gemm-tuning", a.gemm_tuning,
            "--pmc", "off"] + (["--i2v"] if a.i2v else []) + (["--depth", str(a.depth[0]), str(a.depth[1])] if a.depth else [])
        t_pmc = time.perf_counter()
        read, pmc_note = pmc_read_in_this_run(child_argv)
        try:
            pmc_applied = _apply_pmc(read, ps, traffic_per_rate, t_pmc) if read is not None else None
        except Exception as e:      # noqa: BLE001 - never lose the line over the optional leg
            pmc_applied, pmc_note = None, "applying the counters failed: " + repr(e)[:200]
        if pmc_applied is not None:
            traffic, traffic_tbps, traffic_per_rate, traffic_provenance = pmc_applied
        elif read is not None and pmc_note == "ok":
            pmc_note = "the counter passes did not cover every drop rate of the timed launches"
        if pmc_note != "ok":
            traffic_provenance += f" (the in-run counter passes were tried and failed: {pmc_note})"
            print(f"bench.py: in-run counter passes failed: {pmc_note}", file=sys.stderr)
    flops = ps["pairs"] * FLOPS_PER_PAIR
    ach = flops / (ps["total_ms"] * 1e-3) / 1e12 if ps["total_ms"] > 0 else 0.0
    # ---- whole-loop arithmetic: attention FLOPs of the realised lists (this rank's launches) + the dense linear algebra
    #      of the blocks, per video, over `value`
    step_pairs, prev = {}, 0
    for (i, _, _), mk in zip(evs, pair_marks):
        cur = int(mk.item()) if mk is not None else prev
        step_pairs.setdefault(klass(i), []).append(cur - prev)
        prev = cur
    n_div = max(world, sim, 1)
    stage_tokens = [(sh[0] * (sh[1] // 2) * (sh[2] // 2)) for sh in shapes]
    attn_video = gemm_video = 0.0
    for key, n in counts.items():
        if key[1] != "c":
            continue
        pv = step_pairs.get(key) or [v for kk, vv in step_pairs.items() if kk[1] == "c" for v in vv]
        attn_video += n * (sum(pv) / max(len(pv), 1)) * FLOPS_PER_PAIR
        gemm_video += n * hy_gemm_flops_per_computed_step(stage_tokens[key[0]] // n_div, n_txt, len(model.double_blocks),
                                                          len(model.single_blocks))
    loop = {"attention_flops_per_video": attn_video, "gemm_flops_per_video": gemm_video,
            "PFLOPs": round((attn_video + gemm_video) / max(sec_per_video, 1e-9) / 1e15, 4),
            "frac_of_mfma_peak": round((attn_video + gemm_video) / max(sec_per_video, 1e-9) / 1e12 / MFMA_PEAK_TFLOPS, 4),
            "note": "per rank: attention FLOPs of the realised kept lists (4*128^3 per pair) + 2*M*N*K of the blocks' linear "
                    "layers over the computed steps of one video, divided by `value`; dense bf16 MFMA peak 2.5 PFLOP/s"}
    res = {
        "metric": "DiT denoising-loop sec/video (720p,125f,50 steps)",
        "value": round(sec_per_video, 3), "unit": "s/video", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(elapsed * 1e3 / max(len(plan), 1), 3), "higher_is_better": False, "scaling": "strong",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"HunyuanVideo 720x1280x125f Jenga-{a.preset}, 1xMI355X-class GPU per rank: latent "
                               f"{T}x{Hh}x{W}, {len(model.double_blocks)} double + {len(model.single_blocks)} single "
                               "blocks, hidden 3072, 24 heads, S_img=%d S_txt=%d%s" % ((T * Hh * W) // 4, 512 if a.i2v else 256,
                                                                                  " (I2V token_replace)" if a.i2v else ""),
                   "preset": a.preset, "res_rate_list": preset["res"], "step_rate_list": preset["steps"],
                   "scheduler_shift_list": preset["shifts"],
                   "stage_tokens": [(sh[0] * (sh[1] // 2) * (sh[2] // 2)) for sh in shapes],
                   "sa_drop_rates": a.rates, "p_remain_rates": a.p_remain, "valid_text_tokens": a.valid_text,
                   "qk_norm_gain": a.peaky if a.peaky > 0 else 1.0,
                   "latents": (f"smooth field (white noise at 1/{a.coherent:g} resolution, trilinear, + 10 % white noise)"
                               if a.coherent > 0 else "iid N(0,1)"),
                   "schedule": "full 50-step loop" if not sampled else
                   f"sampled steps {plan}; sec/video = sum over (stage, computed|skipped) classes of "
                   f"count x mean step time, counts {dict((f'{k[0]}{k[1]}', n) for k, n in counts.items())}",
                   "ms_per_class": {f"stage{k[0]}_{'computed' if k[1] == 'c' else 'skipped'}": round(mean(v), 2)
                                    for k, v in sorted(cls.items())},
                   "classes_not_sampled": unsampled,     # non-empty only for very small --steps: they borrow a neighbour's mean
                   "parallelism": (f"rank 0 of a simulated ulysses{sim} job on ONE GPU, exchanges replaced by local copies"
                                   + (f" on a side stream behind a delay of {a.sim_exchange_latency_us:g} us + bytes "
                                      f"leaving the rank / {a.sim_exchange_gbps:g} GB/s (the compute stream waits for "
                                      "them as it would for RCCL)" if a.sim_exchange_gbps > 0 else "")
                                   if sim > 1 else "single GPU" if not dist_on else f"ulysses{world} (RCCL all-to-all)"),
                   "weights": "random init N(0,0.02), seed 0", "finite_output": finite,
                   "gemm_selection": (os.path.relpath(gemm_file, ROOT) + (" (RECORDING: not a measurement)"
                                                                          if a.gemm_tuning.startswith("record:") else
                                                                          " (TunableOp replay, tuning off)")
                                      if gemm_file else "hipBLASLt default heuristic")},
        "roofline": {"kernel": attn_kernel_name(), "bound": "mfma", "achieved": round(ach, 1),
                     "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS, 4),
                     "traffic": traffic, "traffic_TBps": traffic_tbps,
                     "traffic_provenance": traffic_provenance,
                     "traffic_source_of_the_committed_constants": f"{pmc_rel}: memory-side bytes per kept block pair from separate rocprofv3 --pmc "
                                       "passes of this kernel at sa-drop 0.7 and 0.8 on one box (FETCH_SIZE / WRITE_SIZE with the "
                                       "guide's gfx950 corrections; lists with little and with much overlap between adjacent "
                                       "query blocks, interpolated at this run's adjacent_shared_frac); traffic = mean over the "
                                       "timed launches; traffic_TBps = traffic / avg_launch_ms (the fabric roof beside the MFMA "
                                       "one: ~8 TB/s)",
                     "traffic_per_rate": traffic_per_rate,
                     "traffic_ratio_to_algorithmic_bytes": None if traffic is None else round(traffic / ATTN_ALGORITHMIC_BYTES, 1),
                     "algorithmic_bytes_per_launch": ATTN_ALGORITHMIC_BYTES,
                     "launches": ps["launches"],
                     "avg_launch_ms": round(ps["total_ms"] / max(ps["launches"], 1), 3),
                     "kept_block_pairs_per_launch": ps["pairs"] // max(ps["launches"], 1),
                     "adjacent_shared_frac": round(ps.get("adjacent_shared_frac", float("nan")), 3),
                     "algorithmic_flops": "4*128^3 per kept (128-query, 128-key) block pair, realised masks"},
    }
    if dist_on or sim > 1:
        from jenga_amd import dit as _dit
        res["config"]["sp_overlap"] = {
            "enabled": _dit.SP_OVERLAP, "mlp_tail_under_o_exchange": _dit.SP_MLP_TAIL,
            "what": "single-stream blocks: Q,K,V exchange posted behind the QKV half of linear1, the MLP half (GEMM + GELU) "
                    "issued under it, its last share under the O exchange; double-stream blocks: Q|K GEMM -> Q,K exchange, "
                    "V GEMM + text stream under it (JENGA_SP_OVERLAP=0: everything in program order as in round 3)"}
    if sim_ex is not None and a.sim_exchange_gbps > 0:
        n_comp = sum(1 for i in plan if i in computed_steps)
        res["config"]["sim_exchange"] = {
            "gbps": a.sim_exchange_gbps, "latency_us": a.sim_exchange_latency_us,
            "simulated_transfer_ms_per_computed_step": round(sim_ex.sim_us_timed / 1e3 / max(n_comp, 1), 2),
            "note": "transfer time posted on the side stream during the timed steps / computed steps; what is NOT hidden "
                    "shows up in value"}
    if rot_ms:
        rs = rot_prof.summary()
        r_ach = rs["pairs"] * FLOPS_PER_PAIR / (rs["total_ms"] * 1e-3) / 1e12 if rs["total_ms"] > 0 else 0.0
        est = sum(n * (rot_ms[key[0]] if key[1] == "c" and key[0] in rot_ms else class_ms(key) if sampled else mean(cls.get(key, [0.0])))
                  for key, n in counts.items()) / 1e3
        res.setdefault("extra", {})["attn_other_kernel"] = {
            "what": f"JENGA_ATTN_FLAGS={other_flags}: the attention kernel that is NOT the default of this run "
                    f"({'LP kernel, csrc/bsattn3.hip' if other_flags & _capi.ATTN_LP else 'pair kernel, csrc/bsattn5.hip'}); "
                    "same box, same lists, not `value`",
            "ms_per_computed_step": {f"stage{k}": round(v, 2) for k, v in rot_ms.items()},
            "s_per_video_estimate": round(est, 3),
            "attention_TFLOPs": round(r_ach, 1), "attention_frac_of_peak": round(r_ach / MFMA_PEAK_TFLOPS, 4),
            "attention_avg_launch_ms": round(rs["total_ms"] / max(rs["launches"], 1), 3),
            "note": "one computed step per stage after the timed region; estimate = the timed run's skipped-step classes + "
                    "these computed-step times x their counts"}
    res["loop"] = loop
    if gemm_choices_synced is not None:
        res["config"]["gemm_choices_synchronized_across_ranks"] = gemm_choices_synced
        mm = torch.tensor([_capi.linear_import_mismatches()], device=dev, dtype=torch.int64)
        if dist_on:
            dist.all_reduce(mm, op=dist.ReduceOp.MAX)
        res["config"]["gemm_choice_solution_index_mismatches_max_over_ranks"] = int(mm.item())
    if xgmi_raw is not None:
        LINK_GBPS = 153.6                      # one xGMI link, one direction (MI355X_MICROARCH.md: 7 links x ~153 GB/s per GPU)
        layers = len(model.double_blocks) + len(model.single_blocks)
        st_recs = []
        for sr in xgmi_raw["stages"]:
            ach = sr["bytes_out_per_layer"] / max(sr["exchange_alone_ms"], 1e-9) / 1e6
            exposed = max(0.0, sr["step_ms"] - sr["step_ms_without_fabric"])
            st_recs.append({"stage": sr["stage"], "S_loc": sr["S_loc"], "bytes_out_per_rank_per_layer": sr["bytes_out_per_layer"],
                            "exchange_alone_ms_per_layer": round(sr["exchange_alone_ms"], 4),
                            "achieved_GBps_per_rank": round(ach, 1),
                            "frac_of_peak": round(ach / max((world - 1) * LINK_GBPS, 1e-9), 4) if world > 1 else None,
                            "computed_step_ms": round(sr["step_ms"], 2),
                            "computed_step_ms_without_fabric": round(sr["step_ms_without_fabric"], 2),
                            "exposed_ms_per_computed_step": round(exposed, 2),
                            "hidden_frac": round(1.0 - exposed / max(layers * sr["exchange_alone_ms"], 1e-9), 4)})
        n1 = xgmi_raw["n1"]
        n1_spv = None
        if n1:
            def n1_ms(key):
                if key in n1:
                    return n1[key]
                same = [v for kk, v in n1.items() if kk[1] == key[1]]
                return same[-1] if same else 0.0
            n1_spv = sum(n * n1_ms(key) for key, n in counts.items()) / 1e3
        res["roofline_xgmi"] = {
            "bound": "xgmi", "ranks": world, "backend": "nccl (RCCL)", "exchange_mode": xgmi_raw["mode"],
            "exchange": "per layer: Q|K all-to-all, V all-to-all, O all-to-all, text all-gather (jenga_amd/modules/ulysses.py; "
                        "reference xdit_ring_atten.py:118-131, 212-217)",
            "algorithmic_bytes": "bytes leaving a rank per layer = 4 x (N-1)/N x S_loc x H x 128 x 2 (Q, K, V, O) + (N-1) x "
                                 "S_txt x H/N x 128 x 2 (text rows of the rank's heads to every peer)",
            "peak_GBps_per_rank": round(max(world - 1, 0) * LINK_GBPS, 1), "peak_source": "(N-1) links x 153.6 GB/s, one direction "
                                  "(MI355X_MICROARCH.md: 7 x ~153 GB/s per GPU; a full mesh, every peer message rides its own link)",
            "unit": "GB/s", "stages": st_recs,
            "how": "exchange_alone: the layer's four collectives with buffers of the layer's shapes, nothing else on the GPU, 20 "
                   "repetitions between two HIP events (the compute stream only waits for RCCL's stream, so the elapsed time "
                   "IS the time on RCCL's stream); exposed: one computed step with the real exchange minus the same step with "
                   "every transfer replaced by a local copy (NoFabricExchange), max over ranks; hidden_frac = 1 - exposed / "
                   "(layers x exchange_alone)",
            "bytes_out_counted_rank0_all_steps": xgmi_raw["bytes_counted"], "exchange_calls_counted_rank0": xgmi_raw["exchange_calls_counted"],
            "n1_same_session": None if n1_spv is None else {
                "s_per_video": round(n1_spv, 3),
                "how": "rank 0, the other ranks idle: the single-rank model (all heads, whole sequence) on the same box, one "
                       "computed and one skipped step per stage x their counts in the 50-step schedule"},
            "efficiency": None if n1_spv is None else round(n1_spv / max(world * sec_per_video, 1e-9), 4),
            "efficiency_note": "T_1 / (N x T_N) with T_1 from n1_same_session; the driver computes its own from its per-N runs"}
    if power_rec is not None:
        res["power"] = power_rec
    if dense_ms is not None:
        res["dense_reference"] = {
            "s_per_video": round(50 * dense_ms / 1e3, 2), "ms_per_dense_step": round(dense_ms, 1),
            "speedup_vs_dense": round(50 * dense_ms / 1e3 / max(sec_per_video, 1e-9), 3),
            "note": "one computed step at sa-drop 0.0 (dense branch of the blocks, same kernels, every kv block kept), "
                    "final resolution, measured after the timed region; the dense loop computes all 50 steps "
                    "(no step skipping).  The reference's own ratio on H800: 1625 s / 310 s = 5.24 (README.md:80-82)"}
    n_layers = len(model.double_blocks) + len(model.single_blocks)
    if rank == 0 and not dist_on and sim <= 1 and not a.no_secondary:
        st = stages[-1]
        try:
            res["roofline_secondary"] = secondary_roofline(
                dev, S_img=st["h2l"].numel(), S_txt=n_txt, top_k=int((1 - a.rates[0]) * (st["h2l"].numel() // 128)),
                p_remain=a.p_remain, nbm=st["curve"][0][2])
        except Exception as e:      # noqa: BLE001
            extras_failed["roofline_secondary"] = repr(e)[:300]
    if rank == 0 and not dist_on and sim <= 1 and not a.no_wan_extra and a.preset == "base" and not a.depth:
        del model
        torch.cuda.empty_cache()
        try:
            res.setdefault("extra", {})["wan14b"] = wan_extra(dev)
        except Exception as e:      # noqa: BLE001
            extras_failed["wan14b"] = repr(e)[:300]
    cb = None
    if rank == 0 and not dist_on and not a.no_cpu_baseline:
        try:
            cb = cpu_baseline(a.rates, a.p_remain)
        except Exception as e:      # noqa: BLE001
            extras_failed["cpu_baseline"] = repr(e)[:300]
    if cb is not None:
        layers = n_layers
        res["cpu_baseline"] = {
            "value": round(cb["s_per_layer"] * layers * len(computed_steps), 1), "unit": "s/video",
            "cores": cb["cores"], "kind": "port", "cpu_model": cb["cpu_model"], "logical_cpus": cb["logical"],
            "sample": "reference PyTorch-CPU eager path restated in torch (oracle/eager_torch.py: torch block selection "
                      "incl. the Hilbert block-neighbour matrix + F.scaled_dot_product_attention with the block mask expanded per 128x128 tile; the reference's "
                      "only CPU-capable attention mode, attenion.py:102-109), torch.set_num_threads(physical cores); "
                      "time-capped samples, extrapolated linearly in query rows: A = 1 head x full S=115456 in fp32 and "
                      f"bf16, B = one 24-head layer of the 0.5-res stage (S=28416) in bf16; value = 24 heads x "
                      f"({cb['best']} leg A) x {layers} layers x {len(computed_steps)} computed steps, attention + "
                      "selection only (GEMMs, norms and RoPE excluded)",
            "detail": cb["detail"]}
    if extras_failed:
        res["extras_failed"] = extras_failed
    # The JSON line has to be the LAST thing on stdout.  RCCL writes its five-line version banner ("RCCL version : ...") to
    # stdout through C stdio when the communicator is created; with stdout a pipe or a file it sits in libc's buffer until the
    # process exits -- i.e. it used to land AFTER the line below in every run that initialises RCCL (seen in
    # JENGA_BENCH_FORCE_DIST=1 runs; an N > 1 run would have done the same to the driver's parser).  So: tear the process group
    # down first, flush libc's buffers, then print.
    _flush_c_stdio()                # every rank's banner goes out now ...
    if dist_on or sim > 1:
        try:
            if dist_on:
                dist.barrier()      # ... and every rank has done so before rank 0 prints
            dist.destroy_process_group()
        except Exception as e:      # noqa: BLE001 - the measurement is complete; say so and still print it
            print(f"bench.py: rank {rank}: destroy_process_group failed ({e!r})", file=sys.stderr)
    _flush_c_stdio()
    if rank == 0:
        sys.stdout.write(json.dumps(res) + "\n")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
