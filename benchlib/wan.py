"""Part of bench.py (repo root): the Wan2.1-14B workload (BASELINE.json configs[3]): --workload wan14b and the `extra.wan14b` estimate.  Split out of bench.py in round 6;
bench.py re-exports these names."""
import json
import os
import sys
import time

import torch

from .consts import ATTN_ALGORITHMIC_BYTES, FLOPS_PER_PAIR, HBM_PEAK_GBPS, MFMA_PEAK_TFLOPS, PMC_FILE, ROOT, attn_kernel_name  # noqa: F401


from .cpu_ref import cpu_baseline
from .power import PowerSampler


# ------------------------------------------------------------------------------------------------ Wan2.1 (configs[3])
WAN_RATE_PRIORITY = [0.7, 0.8, 0.0, 0.571429, 0.285714, 0.428571, 0.142857]   # which step classes a short run samples first


def wan_setup(dev, task="t2v-14B", size=(1280, 720), frames=81, qk_gain=4.0, p_remain=0.8, layers=None):
    """Synthetic-weight Wan2.1 DiT of the real architecture + latents / text of scripts/wan_14B_jenga_base.sh's shape."""
    from jenga_amd import gilbert as G
    from jenga_amd.wan_dit import WAN_CONFIGS, WanDiT
    W, Hh = size
    F_lat, H_lat, W_lat = (frames - 1) // 4 + 1, Hh // 8, W // 8
    grid = (F_lat, H_lat // 2, W_lat // 2)
    L = grid[0] * grid[1] * grid[2]
    cfg = dict(WAN_CONFIGS[task])
    if layers:
        cfg["num_layers"] = layers
    torch.manual_seed(0)
    m = WanDiT(dtype=torch.bfloat16, device=dev, **cfg)
    for p_ in m.parameters():
        if p_.dim() >= 2:
            torch.nn.init.normal_(p_, std=0.02)
    if qk_gain != 1.0:      # random weights give flat pooled scores (the p-remain 0.8 rule would keep ~80 % of the blocks
        for blk in m.blocks:    # at every drop rate); a gain of 4 makes the block softmax as peaked as a trained model's
            blk.self_attn.norm_q.weight.data.mul_(qk_gain)
            blk.self_attn.norm_k.weight.data.mul_(qk_gain)
    l2h, h2l = G.sliced_gilbert_mapping(*grid, as_tensor=True, device=dev)
    nbm = G.sliced_gilbert_block_neighbor_mapping(*grid, as_tensor=True, device=dev)
    m.set_curve(l2h, h2l, nbm)
    m.p_remain_rates = p_remain
    g = torch.Generator(device=dev).manual_seed(42)
    x = [torch.randn(16, F_lat, H_lat, W_lat, generator=g, device=dev)]
    ctx = [torch.randn(100, 4096, generator=g, device=dev)]
    m.enable_teacache(50, 0.15, task, use_ret_steps=True, enable=False)
    return m, x, ctx, L, grid, cfg


def wan_gemm_flops_per_forward(L, dim, ffn, layers, text_len=512):
    """self-attention q,k,v,o + cross-attention q,o (L rows) and k,v (text rows) + ffn, per forward."""
    per = 2.0 * L * dim * dim * 6 + 2.0 * text_len * dim * dim * 2 + 2.0 * L * dim * ffn * 2
    return layers * per


def wan_main(a, dev):
    """--workload wan14b: BASELINE.json configs[3] as a standard line.  A step = one scheduler step = two CFG forwards
    at the step's drop rate (jenga_wan.py:190-206 warm-up ramp, sa-drop 0.7 / 0.8, p-remain 0.8, dense <= 0.25); every
    forward computed (TeaCache's polynomial is calibrated on the trained time embedding, so its skip count on random
    weights is not meaningful -- the replayed count is reported beside the value)."""
    from jenga_amd import _capi
    from jenga_amd.prores import FlowMatchSchedule
    from jenga_amd.wan_driver import sa_drop_rate_for_step
    rates2 = a.rates or [0.7, 0.8]
    p_remain = a.p_remain if a.p_remain is not None else 0.8
    layers = a.depth[0] if a.depth else None
    m, x, ctx, L, grid, cfg = wan_setup(dev, p_remain=p_remain, layers=layers)
    sched = FlowMatchSchedule(50, shift=8.0)
    rates = [round(sa_drop_rate_for_step(i, 50, rates2), 6) for i in range(50)]
    counts = {}
    for r in rates:
        counts[r] = counts.get(r, 0) + 1
    if a.steps >= 50:
        plan, sampled = [rates[i % 50] for i in range(a.steps)], False
    else:
        prio = [r for r in (round(v, 6) for v in WAN_RATE_PRIORITY) if r in counts] + \
               [r for r in sorted(counts) if r not in [round(v, 6) for v in WAN_RATE_PRIORITY]]
        plan, sampled = [prio[j % len(prio)] for j in range(a.steps)], True

    def step(rate):
        for _ in range(2):      # conditional + unconditional stream (jenga_wan.py t2v_generate)
            y = m(x, sched.timesteps[:1].to(dev), ctx, seq_len=L, sa_drop_rate=rate)[0]
        return y

    for w in range(a.warmup):
        step(rates2[w % 2])
    torch.cuda.synchronize()
    _capi.ATTN_PROFILE = prof = _capi.AttnProfile()
    power = PowerSampler().start()
    evs = []
    t0 = time.perf_counter()
    for r in plan:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = step(r)
        e1.record()
        evs.append((r, e0, e1))
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    pw = power.stop()
    _capi.ATTN_PROFILE = None
    finite = bool(torch.isfinite(y).all().item())
    cls = {}
    for r, e0, e1 in evs:
        cls.setdefault(r, []).append(e0.elapsed_time(e1))
    mean = lambda v: sum(v) / len(v)
    if sampled:
        def class_ms(r):        # an unsampled ramp class takes the nearest sampled LOWER rate (slower: conservative)
            if r in cls:
                return mean(cls[r])
            lower = [q_ for q_ in cls if q_ <= r]
            return mean(cls[max(lower)] if lower else cls[min(cls)])
        sec = sum(n * class_ms(r) for r, n in counts.items()) / 1e3
        unsampled = sorted(r for r in counts if r not in cls)
    else:
        sec, unsampled = elapsed * 50.0 / len(plan), []
    ps = prof.summary()
    flops = ps["pairs"] * FLOPS_PER_PAIR
    ach = flops / (ps["total_ms"] * 1e-3) / 1e12 if ps["total_ms"] > 0 else 0.0
    # loop arithmetic over the timed steps: attention FLOPs of the realised lists + the dense linear algebra
    gemm = wan_gemm_flops_per_forward(L, cfg["dim"], cfg["ffn_dim"], cfg["num_layers"]) * 2 * len(plan)
    cross = 4.0 * L * 512 * cfg["dim"] * cfg["num_layers"] * 2 * len(plan)
    res = {
        "metric": "DiT denoising-loop sec/video (Wan2.1-14B 720p, 81f, 50 steps x 2 CFG forwards)",
        "value": round(sec, 3), "unit": "s/video", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(elapsed * 1e3 / max(len(plan), 1), 3), "higher_is_better": False, "scaling": "strong",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"Wan2.1-14B T2V 1280x720x81f Jenga-Base on 1xMI355X (BASELINE.json configs[3], "
                               f"scripts/wan_14B_jenga_base.sh): {L} tokens = {-(-L // 128)} blocks of 128, "
                               f"{cfg['num_layers']} layers, dim {cfg['dim']}, {cfg['num_heads']} heads, ffn {cfg['ffn_dim']}, "
                               "text context 512, weights resident (no offload)",
                   "sa_drop_rates": rates2, "p_remain_rates": p_remain, "first_frame_blocks": -(-L // 128) // 21,
                   "drop_rate_schedule": {str(r): n for r, n in sorted(counts.items())},
                   "schedule": "full 50-step loop" if not sampled else
                   f"sampled step classes {plan} (a step = two CFG forwards at that drop rate); sec/video = sum over the "
                   "drop-rate classes of count x mean step time",
                   "ms_per_step_by_drop_rate": {str(r): round(mean(v), 1) for r, v in sorted(cls.items())},
                   "classes_not_sampled": unsampled, "teacache": "off: every forward computed",
                   "qk_norm_gain": 4.0, "weights": "random init N(0,0.02), seed 0 (norm_q / norm_k weights x 4: peaked "
                                                   "block softmax, top_k decides as in a trained model)",
                   "finite_output": finite, "parallelism": "single GPU"},
        "roofline": {"kernel": attn_kernel_name(), "bound": "mfma", "achieved": round(ach, 1),
                     "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "traffic": None,
                     "launches": ps["launches"], "avg_launch_ms": round(ps["total_ms"] / max(ps["launches"], 1), 3),
                     "kept_block_pairs_per_launch": ps["pairs"] // max(ps["launches"], 1),
                     "algorithmic_flops": "4*128^3 per kept (128-query, 128-key) block pair, realised lists at 591 blocks x "
                                          "40 heads (dense-branch launches of the warm-up ramp included)"},
        "loop": {"flops_timed_steps": flops + gemm + cross, "attention_flops": flops, "gemm_flops": gemm + cross,
                 "PFLOPs": round((flops + gemm + cross) / max(elapsed, 1e-9) / 1e15, 4),
                 "frac_of_mfma_peak": round((flops + gemm + cross) / max(elapsed, 1e-9) / 1e12 / MFMA_PEAK_TFLOPS, 4)},
        "power": pw,
    }
    if not a.no_cpu_baseline:
        cb = cpu_baseline(rates2, p_remain, workload="wan14b")
        res["cpu_baseline"] = {
            "value": round(cb["s_per_layer"] * cfg["num_layers"] * 100, 1), "unit": "s/video", "cores": cb["cores"],
            "kind": "port", "cpu_model": cb["cpu_model"], "logical_cpus": cb["logical"],
            "sample": "reference PyTorch-CPU eager path restated in torch (oracle/eager_torch.py), one head x S = 75648 "
                      f"(591 blocks, first_frame_blocks 28, sliced-Gilbert neighbours) in fp32 and bf16, time-capped and "
                      f"extrapolated linearly in query rows; value = 40 heads x ({cb['best']} leg) x {cfg['num_layers']} layers "
                      "x 100 forwards, self-attention + selection only",
            "detail": cb["detail"]}
    print(json.dumps(res))


def wan_extra(dev):
    """Short configs[3] leg of the default N=1 run (after the timed region): one warm-up + one timed Jenga forward of
    the full Wan2.1-14B model at each of the two drop rates; sec/video as if all 100 forwards ran at those rates."""
    from jenga_amd import _capi
    from jenga_amd.prores import FlowMatchSchedule
    m, x, ctx, L, grid, cfg = wan_setup(dev)
    sched = FlowMatchSchedule(50, shift=8.0)
    t = sched.timesteps[:1].to(dev)
    m(x, t, ctx, seq_len=L, sa_drop_rate=0.8)
    out = {}
    for r in (0.7, 0.8):
        torch.cuda.synchronize()
        _capi.ATTN_PROFILE = prof = _capi.AttnProfile()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        m(x, t, ctx, seq_len=L, sa_drop_rate=r)
        e1.record()
        torch.cuda.synchronize()
        _capi.ATTN_PROFILE = None
        ps = prof.summary()
        ach = ps["pairs"] * FLOPS_PER_PAIR / (ps["total_ms"] * 1e-3) / 1e12
        out[str(r)] = {"ms_per_forward": round(e0.elapsed_time(e1), 1), "attention_TFLOPs": round(ach, 1),
                       "attention_frac_of_peak": round(ach / MFMA_PEAK_TFLOPS, 4),
                       "attention_avg_launch_ms": round(ps["total_ms"] / max(ps["launches"], 1), 3),
                       "kept_block_pairs_per_launch": ps["pairs"] // max(ps["launches"], 1)}
    # 50 steps x 2 forwards: steps 0-4 ramp up (counted at rate[0], slightly optimistic), 5-25 rate[0], 26-49 rate[1]
    est = (2 * 26 * out["0.7"]["ms_per_forward"] + 2 * 24 * out["0.8"]["ms_per_forward"]) / 1e3
    del m
    torch.cuda.empty_cache()
    return {"workload": f"Wan2.1-14B T2V 1280x720x81f Jenga-Base (BASELINE.json configs[3]): {L} tokens, 40 layers, dim 5120, "
                        "40 heads, p-remain 0.8, qk-norm gain 4; `python bench.py --workload wan14b` is the full line",
            "per_drop_rate": out, "s_per_video_two_rate_estimate": round(est, 1),
            "note": "one timed forward per drop rate after one warm-up forward; estimate = 52 forwards at 0.7 + 48 at 0.8 "
                    "(the five ramp steps of jenga_wan.py:205-206 counted at 0.7)"}
