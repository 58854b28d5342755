"""Part of bench.py (repo root): the cpu_baseline leg (oracle/eager_torch.py timed on the host cores; the only place bench.py touches oracle/).  Split out of bench.py in round 6;
bench.py re-exports these names."""
import json
import os
import sys
import time

import torch

from .consts import ATTN_ALGORITHMIC_BYTES, FLOPS_PER_PAIR, HBM_PEAK_GBPS, MFMA_PEAK_TFLOPS, PMC_FILE, ROOT  # noqa: F401


from .power import _host_cpu


def cpu_baseline(rates, p_remain, budget_s=9.0, workload="hy720p"):
    """The reference's PyTorch-CPU eager path (restated in oracle/eager_torch.py: its torch block
    selection + F.scaled_dot_product_attention with the block mask expanded per 128x128 tile), timed on this box's
    host cores on BOUNDED samples and extrapolated linearly, attention + selection only (GEMMs excluded):
      hy720p  A  1 head x the full 720p sequence (S = 115456), fp32 and bf16: as many query-block chunks as fit the budget;
              B  one layer of the 0.5-resolution stage (24 heads, S = 28416), bf16: as many heads as fit the budget.
      wan14b  A  1 head x the 1280x720x81f sequence (S = 75648 = 591 blocks, no text blocks, sliced-Gilbert neighbours,
                 first_frame_blocks = 28), fp32 and bf16."""
    from oracle import eager_torch as et
    from oracle import gilbert as og
    model, physical, logical = _host_cpu()
    torch.set_num_threads(physical)
    out = {}
    gen = torch.Generator().manual_seed(1)

    def one(S_img_blocks, grid, heads, dtype, budget, tb=2, sliced=False, ffb=0, valid_text=64):
        nb = S_img_blocks + tb
        S = nb * 128
        # the static Hilbert block adjacency (the oracle's C Gilbert)
        nbm = (og.sliced_gilbert_block_neighbor_mapping if sliced else og.gilbert_block_neighbor_mapping)(*grid)
        q = torch.randn(1, heads, S, 128, generator=gen).to(dtype)
        k = torch.randn(1, heads, S, 128, generator=gen).to(dtype)
        v = torch.randn(1, heads, S, 128, generator=gen).to(dtype)
        top_k = int((1 - rates[0]) * S_img_blocks)
        t0 = time.perf_counter()
        mask = et.build_block_mask(q[:, :, : S_img_blocks * 128], k, top_k, S_img_blocks, nb, p_remain, tb, nbm,
                                   first_frame_blocks=ffb)
        t_sel = time.perf_counter() - t0
        t0 = time.perf_counter()
        _, frac = et.masked_attention(q, k, v, mask, S_img_blocks * 128 + (valid_text if tb else 0), S_img_blocks,
                                      q_chunk_blocks=16, budget_s=budget, clock=time.perf_counter)
        t_att = time.perf_counter() - t0
        return t_sel, t_att / max(frac, 1e-9), frac, float(mask.float().mean())

    legs = []
    if workload == "wan14b":
        heads_total, shape = 40, dict(S_img_blocks=591, grid=(21, 45, 80), tb=0, sliced=True, ffb=28)
        tag = "wan720p_1head"
    else:
        heads_total, shape = 24, dict(S_img_blocks=900, grid=(32, 45, 80))
        tag = "full_res_1head"
    for name, dtype in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        t_sel, t_att, frac, dens = one(heads=1, dtype=dtype, budget=budget_s, **shape)
        legs.append((name, t_sel, t_att, frac, dens))
        out[f"{tag}_{name}"] = {"selection_s": round(t_sel, 3), "attention_s_extrapolated": round(t_att, 2),
                                "rows_timed_frac": round(frac, 4), "kept_block_frac": round(dens, 3)}
    if workload != "wan14b":
        t_sel, t_att, frac, dens = one(220, (32, 22, 40), 24, torch.bfloat16, budget_s)
        out["half_res_layer_24heads_bf16"] = {"selection_s": round(t_sel, 3), "attention_s_extrapolated": round(t_att, 2),
                                              "rows_timed_frac": round(frac, 4), "kept_block_frac": round(dens, 3)}
    best = min(legs, key=lambda l_: l_[1] + l_[2])
    per_layer = heads_total * (best[1] + best[2])          # all heads, one AttenCarve call
    return dict(s_per_layer=per_layer, best=best[0], detail=out, cpu_model=model, cores=physical, logical=logical)
