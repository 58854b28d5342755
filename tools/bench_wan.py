"""Wan2.1 denoising-loop timing on one GPU (BASELINE.json configs[3]: Wan2.1-14B T2V 720p Jenga-Base, no offload).

    python tools/bench_wan.py --task t2v-14B --size 1280x720 --frames 81

Synthetic latents / text / random weights of the real architecture (jenga_amd.wan_dit.WanDiT), the reference's schedule
(scripts/wan_14B_jenga_base.sh: 50 steps, two CFG forwards per step, sa-drop 0.7 / 0.8 with the warm-up ramp of
jenga_wan.py:190-206, p-remain 0.8, dense below 0.25).  One forward is timed per distinct drop rate and the loop time
is the count-weighted sum; the TeaCache line replays the reference's skip rule (thresh 0.15, ret-steps) on this
model's time embeddings -- with random weights the rule's polynomial is not calibrated, so the number of computed
forwards is reported next to the time.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jenga_amd import gilbert as G  # noqa: E402
from jenga_amd.prores import FlowMatchSchedule  # noqa: E402
from jenga_amd.wan_dit import WAN_CONFIGS, WanDiT  # noqa: E402
from jenga_amd.wan_driver import TeaCache, sa_drop_rate_for_step  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--task", default="t2v-14B", choices=list(WAN_CONFIGS))
    ap.add_argument("--size", default="1280x720")
    ap.add_argument("--frames", type=int, default=81)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--rates", type=float, nargs="+", default=[0.7, 0.8])
    ap.add_argument("--p-remain", type=float, default=0.8)
    ap.add_argument("--thresh", type=float, default=0.15)
    ap.add_argument("--shift", type=float, default=8.0)
    ap.add_argument("--qk-gain", type=float, default=1.0,
                    help="multiplies the self-attention norm_q / norm_k weights: random weights give flat pooled scores, "
                         "so the p-remain rule (0.8 of the probability mass) keeps ~80 %% of the blocks at every drop rate; "
                         "a gain of 3-4 makes the block softmax as peaked as a trained model's and top_k decides")
    ap.add_argument("--layers", type=int, default=None, help="debug: fewer layers")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    W, H = (int(v) for v in a.size.lower().replace("*", "x").split("x"))
    F_lat, H_lat, W_lat = (a.frames - 1) // 4 + 1, H // 8, W // 8
    grid = (F_lat, H_lat // 2, W_lat // 2)
    L = grid[0] * grid[1] * grid[2]
    cfg = dict(WAN_CONFIGS[a.task])
    if a.layers:
        cfg["num_layers"] = a.layers
    torch.manual_seed(0)
    t0 = time.time()
    m = WanDiT(dtype=torch.bfloat16, device=dev, **cfg)
    for p_ in m.parameters():
        if p_.dim() >= 2:
            torch.nn.init.normal_(p_, std=0.02)
    if a.qk_gain != 1.0:
        for blk in m.blocks:
            blk.self_attn.norm_q.weight.data.mul_(a.qk_gain)
            blk.self_attn.norm_k.weight.data.mul_(a.qk_gain)
    l2h, h2l = G.sliced_gilbert_mapping(*grid, as_tensor=True)
    nbm = G.sliced_gilbert_block_neighbor_mapping(*grid, as_tensor=True)
    m.set_curve(l2h, h2l, nbm)
    m.p_remain_rates = a.p_remain
    g = torch.Generator(device=dev).manual_seed(42)
    x = [torch.randn(16, F_lat, H_lat, W_lat, generator=g, device=dev)]
    ctx = [torch.randn(100, 4096, generator=g, device=dev)]
    sched = FlowMatchSchedule(a.steps, shift=a.shift)
    rates = [sa_drop_rate_for_step(i, a.steps, a.rates) for i in range(a.steps)]
    classes = sorted(set(round(r, 6) for r in rates))
    torch.cuda.synchronize()
    setup_s = time.time() - t0
    # --- one timed forward per distinct drop rate (teacache off), after one warm-up forward
    m.enable_teacache(a.steps, a.thresh, a.task, use_ret_steps=True, enable=False)
    m(x, sched.timesteps[:1].to(dev), ctx, seq_len=L, sa_drop_rate=classes[-1])
    ms = {}
    for r in classes:
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = m(x, sched.timesteps[:1].to(dev), ctx, seq_len=L, sa_drop_rate=r)[0]
        e1.record()
        torch.cuda.synchronize()
        ms[r] = e0.elapsed_time(e1)
    finite = bool(torch.isfinite(y).all().item())
    full = sum(2 * ms[round(r, 6)] for r in rates) / 1e3
    # --- TeaCache replay on the time embeddings only (no block compute)
    tea = TeaCache(a.steps, a.thresh, a.task, use_ret_steps=True, enable=True)
    computed, tea_s, skip_ms = 0, 0.0, 0.0
    for i in range(a.steps):
        t = sched.timesteps[i:i + 1].to(dev)
        from jenga_amd.wan_dit import sinusoidal_embedding_1d
        e = m.time_embedding(sinusoidal_embedding_1d(m.freq_dim, t).float())
        e0_ = m.time_projection(e).unflatten(1, (6, m.dim))
        for _ in range(2):
            calc, _par = tea.decide(e, e0_)
            tea.advance()
            computed += int(calc)
            tea_s += ms[round(rates[i], 6)] / 1e3 if calc else 0.0
    out = {"metric": "Wan2.1 DiT denoising-loop sec/video", "task": a.task, "size": a.size, "frames": a.frames,
           "steps": a.steps, "forwards_per_step": 2, "tokens": L, "blocks_128": -(-L // 128),
           "value_all_computed": round(full, 2), "value_teacache_replay": round(tea_s, 2),
           "teacache_computed_forwards": computed, "unit": "s/video", "n_gpus": 1, "dtype": "bf16",
           "data": "synthetic", "sa_drop_rates": a.rates, "p_remain_rates": a.p_remain,
           "ms_per_forward_by_drop_rate": {str(k): round(v, 1) for k, v in ms.items()},
           "steps_per_drop_rate": {str(c): sum(1 for r in rates if round(r, 6) == c) for c in classes},
           "qk_gain": a.qk_gain, "layers": cfg["num_layers"], "dim": cfg["dim"], "heads": cfg["num_heads"], "setup_s": round(setup_s, 1),
           "finite_output": finite}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
