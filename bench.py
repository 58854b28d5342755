#!/usr/bin/env python3
"""Benchmark of the north-star metric: HunyuanVideo DiT denoising loop, sec/video (720x1280, 125 frames, 50 steps)
at Jenga-Base settings, on N GPUs of one node (Ulysses sequence parallelism over RCCL for N > 1).

    python bench.py --gpus 1 --steps 6 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

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

MFMA_PEAK_TFLOPS = 2500.0  # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
FLOPS_PER_PAIR = 4 * 128 ** 3  # one 128x128 query block against one 128-key block, head_dim 128: QK^T + PV


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rates", type=float, nargs=2, default=[0.75, 0.85],
                    help="sa-drop-rates per stage; 0.75 0.85 = the shipped Jenga-Base script, 0.7 0.8 = BASELINE.json's pair")
    ap.add_argument("--p-remain", type=float, default=0.3)
    ap.add_argument("--latent", type=int, nargs=3, default=[32, 90, 160], help="latent T H W (720x1280x125f)")
    ap.add_argument("--depth", type=int, nargs=2, default=None, help="override (double, single) block counts (debug only)")
    ap.add_argument("--valid-text", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def rate_for_step(i, rates):
    return rates[0] if i <= 25 else rates[1]     # step-rate-list 0.5 1.0 (pipeline...prores.py:422,697-698)


def cpu_baseline(rates, p_remain):
    """Oracle AttenCarve op on the host cores, bounded sample, extrapolated by kept block pairs to one video."""
    import numpy as np
    from oracle import attention as oa
    from oracle import gilbert as og
    H, nimg, tb = 1, 40, 2                     # 5120 image tokens + 256 text tokens, one head
    grid = (5, 16, 64)
    nbm = og.gilbert_block_neighbor_mapping(*grid, 128)
    gen = torch.Generator().manual_seed(1)
    S = (nimg + tb) * 128
    q = torch.randn(1, S, H, 128, generator=gen).to(torch.bfloat16).float().numpy()
    k = torch.randn(1, S, H, 128, generator=gen).to(torch.bfloat16).float().numpy()
    v = torch.randn(1, S, H, 128, generator=gen).to(torch.bfloat16).float().numpy()
    cu = np.array([0, nimg * 128 + 64, S], np.int32)
    top_k = int((1 - rates[0]) * nimg)
    t0 = time.time()
    reps = 0
    pairs = 0
    while time.time() - t0 < 12.0:
        o, mask = oa.block_sparse_attention(q, k, v, top_k, "bfloat16", cu_seqlens_q=cu, text_blocks=tb,
                                            block_neighbor_list=nbm, p_remain_rates=p_remain, return_mask=True)
        pairs += int(mask.sum()) + H * tb * (nimg + tb)
        reps += 1
    dt = time.time() - t0
    return dict(pairs_per_s=pairs / dt, cores=os.cpu_count(),
                sample=f"oracle (numpy) block_sparse_attention, bf16 rounding points, 1 head x {nimg}+{tb} blocks "
                       f"(S={S}), {reps} reps in {dt:.1f} s; extrapolated by kept block pairs to 60 layers x 23 "
                       f"computed steps; attention only (GEMMs excluded)")


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    from jenga_amd import _capi
    from jenga_amd.dit import NON_SKIP_STEPS, JengaHYVideoDiT
    from jenga_amd.modules import ulysses

    torch.manual_seed(0)
    kw = {}
    if a.depth:
        kw = dict(depth_double=a.depth[0], depth_single=a.depth[1])
    model = JengaHYVideoDiT(dtype=torch.bfloat16, device=dev, **kw).init_synthetic_weights(0.02, seed=0)
    if world > 1:
        ulysses.init_sequence_parallel()
        for blk in list(model.double_blocks) + list(model.single_blocks):
            blk.hybrid_seq_parallel_attn = ulysses.UlyssesAttenCarve()
    T, Hh, W = a.latent
    cos, sin = model.set_stage((T, Hh, W), dev)
    g = torch.Generator(device=dev).manual_seed(42)
    latents = torch.randn(1, 16, T, Hh, W, generator=g, device=dev, dtype=torch.bfloat16)
    g2 = torch.Generator(device=dev).manual_seed(43)
    text = torch.randn(1, 256, 4096, generator=g2, device=dev, dtype=torch.bfloat16)
    text2 = torch.randn(1, 768, generator=g2, device=dev, dtype=torch.bfloat16)
    text_mask = torch.zeros(1, 256, dtype=torch.int64, device=dev)
    text_mask[:, : a.valid_text] = 1
    guidance = torch.tensor([6000.0], device=dev)
    model.p_remain_rates = a.p_remain
    model.text_amp = 0.0
    model.num_steps = 50
    model.enable_skip = True

    def run_step(i):
        model.cnt = i
        model.sa_drop_rate = rate_for_step(i, a.rates)
        tval = torch.tensor([1000.0 * (1 - i / 50)], device=dev)
        return model(latents, tval, text_states=text, text_mask=text_mask, text_states_2=text2, freqs_cos=cos,
                     freqs_sin=sin, guidance=guidance, return_dict=False)

    if a.steps >= 50:
        plan = [i % 50 for i in range(a.steps)]      # whole loop(s); sec/video = elapsed * 50 / steps
        sampled = False
    else:
        pattern = [0, 5, 7, 26, 27, 29]        # computed@r0, skipped, computed@r0, computed@r1, skipped, computed@r1
        plan = [pattern[j % len(pattern)] for j in range(a.steps)]
        sampled = True

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for w in range(a.warmup):
        run_step(0 if w % 2 == 0 else 26)       # computed steps: also fills previous_residual
    barrier()
    _capi.ATTN_PROFILE = prof = _capi.AttnProfile()
    evs = []
    t0 = time.perf_counter()
    for i in plan:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = run_step(i)
        e1.record()
        evs.append((i, e0, e1))
    barrier()
    elapsed = time.perf_counter() - t0
    _capi.ATTN_PROFILE = None
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    finite = bool(torch.isfinite(out.float()).all().item())

    cls = {"c0": [], "c1": [], "skip": []}
    for i, e0, e1 in evs:
        ms = e0.elapsed_time(e1)
        key = "skip" if i not in NON_SKIP_STEPS else ("c0" if i <= 25 else "c1")
        cls[key].append(ms)
    mean = lambda v: sum(v) / len(v) if v else float("nan")
    if sampled:
        n_c0 = sum(1 for s in NON_SKIP_STEPS if s <= 25)
        n_c1 = len(NON_SKIP_STEPS) - n_c0
        n_skip = 50 - len(NON_SKIP_STEPS)
        parts = [(n_c0, mean(cls["c0"])), (n_c1, mean(cls["c1"])), (n_skip, mean(cls["skip"]))]
        if world > 1:   # scale the per-class event times so that they sum to the max-over-ranks wall time
            scale = elapsed * 1e3 / max(sum(e0.elapsed_time(e1) for _, e0, e1 in evs), 1e-9)
        else:
            scale = 1.0
        sec_per_video = sum(n * t for n, t in parts if n and t == t) * scale / 1e3
    else:
        sec_per_video = elapsed * 50.0 / len(plan)
    ps = prof.summary()
    traffic = None
    try:   # HBM-side bytes per launch from the committed PMC passes (profiles/), scaled by this run's kept pairs
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_bsattn.json")))
        traffic = int(pmc["derived"]["traffic_bytes_per_kept_pair"] * ps["pairs"] / max(ps["launches"], 1))
    except Exception:
        pass
    flops = ps["pairs"] * FLOPS_PER_PAIR
    ach = flops / (ps["total_ms"] * 1e-3) / 1e12 if ps["total_ms"] > 0 else 0.0
    res = {
        "metric": "DiT denoising-loop sec/video (720p,125f,50 steps)",
        "value": round(sec_per_video, 3), "unit": "s/video", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(elapsed * 1e3 / max(len(plan), 1), 3), "higher_is_better": False, "scaling": "strong",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "HunyuanVideo 720x1280x125f Jenga-Base, 1xMI355X-class GPU per rank: latent "
                               f"{T}x{Hh}x{W}, {len(model.double_blocks)} double + {len(model.single_blocks)} single "
                               "blocks, hidden 3072, 24 heads, S_img=%d S_txt=256" % ((T * Hh * W) // 4),
                   "sa_drop_rates": a.rates, "p_remain_rates": a.p_remain, "valid_text_tokens": a.valid_text,
                   "schedule": "full 50-step loop" if not sampled else
                   f"sampled steps {plan}; sec/video = 12*t(computed@rate0) + 11*t(computed@rate1) + 27*t(skipped)",
                   "ms_computed_rate0": round(mean(cls["c0"]), 2), "ms_computed_rate1": round(mean(cls["c1"]), 2),
                   "ms_skipped": round(mean(cls["skip"]), 2),
                   "parallelism": "single GPU" if world == 1 else f"ulysses{world} (RCCL all-to-all)",
                   "weights": "random init N(0,0.02), seed 0", "finite_output": finite},
        "roofline": {"kernel": "jenga::bsattn_fwd_kernel<bf16>", "bound": "mfma", "achieved": round(ach, 1),
                     "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS, 4),
                     "traffic": traffic,
                     "traffic_source": "profiles/r01_pmc_bsattn.json: (2*FETCH_SIZE + WRITE_SIZE) per kept block pair "
                                       "from separate rocprofv3 --pmc passes, x this run's pairs per launch",
                     "launches": ps["launches"],
                     "avg_launch_ms": round(ps["total_ms"] / max(ps["launches"], 1), 3),
                     "kept_block_pairs_per_launch": ps["pairs"] // max(ps["launches"], 1),
                     "algorithmic_flops": "4*128^3 per kept (128-query, 128-key) block pair, realised masks"},
    }
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cb = cpu_baseline(a.rates, a.p_remain)
        # pairs per video: measured pairs per launch by class -> 60 layers x 23 computed steps
        launches_per_step = len(model.double_blocks) + len(model.single_blocks)
        computed = [i for i in plan if i in NON_SKIP_STEPS]
        if computed and ps["launches"]:
            pairs_per_step = ps["pairs"] / len(computed)
            video_pairs = pairs_per_step * len(NON_SKIP_STEPS)
            res["cpu_baseline"] = {"value": round(video_pairs / cb["pairs_per_s"], 1), "unit": "s/video",
                                   "cores": cb["cores"], "kind": "port", "sample": cb["sample"]}
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
