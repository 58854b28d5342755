"""Part of bench.py (repo root): rooflines of the kernels beside the attention kernel, and the GEMM FLOP model of a computed step.  Split out of bench.py in round 6;
bench.py re-exports these names."""
import json
import os
import sys
import time

import torch

from .consts import ATTN_ALGORITHMIC_BYTES, FLOPS_PER_PAIR, HBM_PEAK_GBPS, MFMA_PEAK_TFLOPS, PMC_FILE, ROOT  # noqa: F401


def _timed(fn, reps=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def secondary_roofline(dev, S_img=115200, S_txt=256, H=24, C=3072, mlp=12288, top_k=270, p_remain=0.3, nbm=None):
    """the HBM-bound kernels of the path against 8 TB/s and the GEMM classes against the MFMA peak, at
    the workload's shapes: HIP events around 10 back-to-back launches on the current stream, AFTER the timed region
    (isolated launches: the in-loop shares are in profiles/*_kernel_stats.csv).  Bytes / FLOPs are ALGORITHMIC."""
    from jenga_amd import _capi
    bf = torch.bfloat16
    S = S_img + S_txt
    nb, nimg = S // 128, S_img // 128
    g = torch.Generator(device=dev).manual_seed(7)
    rnd = lambda *shape: torch.randn(*shape, generator=g, device=dev, dtype=bf)
    out = {}

    def hbm(name, ms, nbytes, note):
        out[name] = {"bound": "hbm", "ms": round(ms, 4), "bytes": int(nbytes), "achieved": round(nbytes / ms / 1e6, 1),
                     "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(nbytes / ms / 1e6 / HBM_PEAK_GBPS, 4),
                     "algorithmic_bytes": note}

    x = rnd(1, S_img, C)
    order = torch.randperm(S_img, generator=g, device=dev)
    ms = _timed(lambda: _capi.gather_rows(x, order))
    hbm("gather_rows", ms, 2 * x.numel() * 2 + S_img * 8, "read + write [1,S_img,3072] bf16 + the int64 index")
    vec = rnd(1, C)
    ms = _timed(lambda: _capi.ln_modulate(x, vec, vec))
    hbm("ln_modulate", ms, 2 * x.numel() * 2, "read + write [1,S_img,3072] bf16")
    del x
    qkv = rnd(1, S, 3, H, 128)
    cos = torch.randn(S_img, 128, generator=g, device=dev)
    sin = torch.randn(S_img, 128, generator=g, device=dev)
    w = torch.ones(128, device=dev, dtype=bf)
    q, k = torch.empty((1, S, H, 128), dtype=bf, device=dev), torch.empty((1, S, H, 128), dtype=bf, device=dev)
    qp, kp = torch.empty((1, H, nimg, 128), dtype=bf, device=dev), torch.empty((1, H, nb, 128), dtype=bf, device=dev)
    ms = _timed(lambda: _capi.qk_norm_rope_pool(qkv[:, :, 0], qkv[:, :, 1], w, w, cos, sin, q, k, s_rope=S_img, qpool=qp,
                                                kpool=kp))
    hbm("qk_norm_rope_pool", ms, 4 * q.numel() * 2 + 2 * cos.numel() * 4 + (qp.numel() + kp.numel()) * 2,
        "read Q,K + fp32 cos,sin tables, write Q,K + pooled Q,K")
    ms = _timed(lambda: _capi.pack_v(qkv[:, :, 2], nb))
    hbm("pack_v", ms, 2 * q.numel() * 2, "read + write V [1,S,24,128] bf16")
    ms = _timed(lambda: _capi.block_select(qp, kp, nbm, nimg, nb - nimg, top_k, p_remain))
    lists = H * nimg * nb * 4 + H * nimg * 4
    out["block_select"] = {"bound": "VALU issue (>= 84 % busy, profiles/r05_pmc_select.json; one workgroup per 4 query blocks of a head: 4 x 900 sequential-fma dot products from "
                                    "LDS-staged pooled K, then one wave per row: softmax, bitonic sort in registers, exact "
                                    "shuffle-scan cumulative sum, compaction; DESIGN.md 3.5)", "ms": round(ms, 4),
                           "bytes": int(lists + (qp.numel() + kp.numel()) * 2),
                           "achieved": round((lists + (qp.numel() + kp.numel()) * 2) / ms / 1e6, 1), "peak": HBM_PEAK_GBPS,
                           "unit": "GB/s", "frac": round((lists + (qp.numel() + kp.numel()) * 2) / ms / 1e6 / HBM_PEAK_GBPS, 4),
                           "algorithmic_bytes": "pooled Q,K in, kept lists idx int32 [24,900,902] + cnt out"}
    del qkv, q, k, cos, sin
    # ---- GEMM classes (hipBLASLt; jenga_linear where an epilogue rides along)
    gem = {}
    xi = rnd(1, S_img, C)

    def gemm(name, M, N, K, fn):
        ms_ = _timed(fn, reps=6, warm=2)
        fl = 2.0 * M * N * K
        gem[name] = {"M": M, "N": N, "K": K, "ms": round(ms_, 3), "achieved": round(fl / ms_ / 1e9, 1),
                     "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(fl / ms_ / 1e9 / MFMA_PEAK_TFLOPS, 4)}

    wq = rnd(3 * C, C) * 0.02
    gemm("qkv (double block, img)", S_img, 3 * C, C, lambda: torch.nn.functional.linear(xi, wq))
    wp, gate, bias = rnd(C, C) * 0.02, torch.randn(C, generator=g, device=dev), rnd(C)
    gemm("proj + gate*y + residual epilogue", S_img, C, C, lambda: _capi.linear(xi, wp, bias, gate=gate, res=xi))
    w1 = rnd(mlp, C) * 0.02
    b1 = rnd(mlp)
    hbuf = torch.empty((1, S_img, mlp), dtype=bf, device=dev)
    gemm("fc1 + tanh-GELU epilogue", S_img, mlp, C, lambda: _capi.linear(xi, w1, b1, act=_capi.ACT_GELU_TANH, out=hbuf))
    w2 = rnd(C, mlp) * 0.02
    gemm("fc2 + gate*y + residual epilogue", S_img, C, mlp, lambda: _capi.linear(hbuf, w2, bias, gate=gate, res=xi))
    del hbuf, w1, w2
    xs = rnd(1, S, C)
    cat = torch.empty((1, S, C + mlp), dtype=bf, device=dev)
    wl = rnd(mlp, C) * 0.02
    gemm("linear1 MLP half + GELU into linear2's concat buffer", S, mlp, C,
         lambda: _capi.linear(xs, wl, None, act=_capi.ACT_GELU_TANH, out=cat[..., C:]))
    w3 = rnd(C, C + mlp) * 0.02
    gemm("linear2 + gate*y + residual epilogue", S, C, C + mlp, lambda: _capi.linear(cat, w3, bias, gate=gate, res=xs))
    out["gemm"] = gem
    out["note"] = ("HBM peak = the nominal 8 TB/s; a plain streaming copy of 2 x 708 MB reaches 6.6 TB/s on this chip (tools/micro/"
                   "hbm_copy.hip, profiles/r04_micro_hbm_copy.txt: read-only 7.2, write-only 5.2).  "
                   "Isolated: 6-10 back-to-back launches per kernel between two HIP events, after the timed region (short "
                   "runs read a few % high against the power-capped steady state of the loop); in-loop shares: "
                   "profiles/r04_bench_default_kernel_stats.csv")
    return out


def hy_gemm_flops_per_computed_step(S_img, S_txt, n_double, n_single, C=3072, mlp=12288):
    """Dense linear algebra of one computed forward (models_mul_block_gc_ha_multigpu.py:852-869 dims): double blocks
    qkv + proj + fc1 + fc2 on both streams, single blocks linear1 + linear2."""
    S = S_img + S_txt
    dbl = 2.0 * S * C * (3 * C + C + 2 * mlp)
    sgl = 2.0 * S * (C * (3 * C + mlp) + (C + mlp) * C)
    return n_double * dbl + n_single * sgl
