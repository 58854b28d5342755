#!/usr/bin/env python3
"""Kernel-level micro-benchmark of the AttenCarve op at the HunyuanVideo 720p x 125f shape
(S_img = 115200 = 900 blocks, S_txt = 256, H = 24, D = 128, bf16).  Prints one JSON line per stage.
  python tools/bench_attn.py [--heads 24] [--drop 0.75] [--p 0.3] [--peaky 0] [--iters 5] [--no-xcd]"""
import argparse
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jenga_amd import _capi, gilbert as G  # noqa: E402


def timed(fn, iters, warmup=1):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--heads", type=int, default=24)
    ap.add_argument("--grid", type=int, nargs=3, default=[32, 45, 80])
    ap.add_argument("--drop", type=float, default=0.75)
    ap.add_argument("--p", type=float, default=0.3)
    ap.add_argument("--peaky", type=float, default=0.0, help=">0: clustered block means with this temperature")
    ap.add_argument("--coherent", type=int, default=0, metavar="SMOOTH",
                    help="SURVEY.md 8(d)'s second regime: K block means = centroid + noise with centroids smoothed SMOOTH "
                         "times over the Hilbert block-neighbour graph (spatially smooth features); Q of block m drawn "
                         "near the mean centroid of its Hilbert neighbours.  Adjacent query blocks then keep similar "
                         "lists, as a trained model's do; --gain sets the temperature so that top_k decides n")
    ap.add_argument("--gain", type=float, default=3.0, help="--coherent: centroid scale (pooled score ~ gain^2)")
    ap.add_argument("--sorted", action="store_true", help="kept-count-aware launch order (ATTN_SORTED)")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--valid-text", type=int, default=64)
    ap.add_argument("--no-xcd", action="store_true")
    ap.add_argument("--same-list", action="store_true", help="every query block keeps the SAME blocks (perfect L2 reuse)")
    ap.add_argument("--attn-only", action="store_true")
    ap.add_argument("--l2-resident", type=int, default=0,
                    help="N>0: every query block reads the same N kv blocks repeatedly (same pair count): isolates MFMA/LDS work from L2-miss fill traffic")
    ap.add_argument("--window", type=int, default=0,
                    help="experiment: every query block keeps a random 31.6 %% of the kv blocks [0, WINDOW) only")
    ap.add_argument("--k-head-major", action="store_true", help="K (and Q) stored [B,H,S,D]: contiguous 16 KiB K tiles")
    ap.add_argument("--flags", type=int, default=None,
                    help="attention flags (_capi.ATTN_*): 1 = XCD remap, 4 = balanced launch, 8 = LP kernel, 16 = kept-count-aware "
                         "order (29 = the LP default), 64 = pair kernel (69 = its default), 0 / 1 = round-1 kernel")
    ap.add_argument("--also-flags", type=int, nargs="*", default=[],
                    help="more flag sets timed on the SAME tensors and lists in the same process (A/B on one box); each is "
                         "reported under also[<flags>] with its max |o - o_first|")
    ap.add_argument("--pair-overlap", type=float, default=-1.0,
                    help=">= 0: synthetic lists -- every odd query block shares this fraction of its list with the even "
                         "block in front of it (what Hilbert-adjacent blocks of a trained model look like); same counts")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    t, h, w = a.grid
    S_img, tb = t * h * w, 2
    assert S_img % 128 == 0
    nimg, nb = S_img // 128, S_img // 128 + tb
    S = nb * 128
    H = a.heads
    torch.manual_seed(0)
    nbm = G.gilbert_block_neighbor_mapping(t, h, w, as_tensor=True)
    q = torch.randn(1, S, H, 128, device=dev, dtype=torch.bfloat16)
    k = torch.randn(1, S, H, 128, device=dev, dtype=torch.bfloat16)
    v = torch.randn(1, S, H, 128, device=dev, dtype=torch.bfloat16)
    if a.peaky > 0:
        cent = torch.randn(1, nb, 1, H, 128, device=dev) * a.peaky
        k = (k.view(1, nb, 128, H, 128).float() + cent).to(torch.bfloat16).view(1, S, H, 128)
        pick = torch.randint(0, nimg, (nb,), device=dev)
        q = (q.view(1, nb, 128, H, 128).float() + cent[:, pick]).to(torch.bfloat16).view(1, S, H, 128)
    if a.coherent > 0:
        g = torch.Generator(device=dev).manual_seed(11)
        cent = torch.randn(nb, H, 128, device=dev, generator=g)
        A = torch.zeros(nb, nb, device=dev)
        A[:nimg, :nimg] = nbm.to(dev).float()
        A[nimg:, nimg:] = torch.eye(tb, device=dev)
        A = A / A.sum(-1, keepdim=True)
        for _ in range(a.coherent):
            cent = torch.einsum("mn,nhd->mhd", A, cent)
            cent = cent / cent.norm(dim=-1, keepdim=True) * (128 ** 0.5)      # keep |c|^2 = 128 like white noise
        qcent = torch.einsum("mn,nhd->mhd", A, cent)                          # mean centroid of the Hilbert neighbours
        qcent = qcent / qcent.norm(dim=-1, keepdim=True) * (128 ** 0.5)
        noise = lambda: torch.randn(1, nb, 128, H, 128, device=dev, generator=g)
        k = (noise() + a.gain * cent.view(1, nb, 1, H, 128)).to(torch.bfloat16).view(1, S, H, 128)
        q = (noise() + a.gain * qcent.view(1, nb, 1, H, 128)).to(torch.bfloat16).view(1, S, H, 128)
    top_k = int((1 - a.drop) * nimg)
    seqlens = torch.tensor([S_img + a.valid_text], dtype=torch.int32, device=dev)

    res = {"shape": dict(S=S, H=H, nb=nb, top_k=top_k, p=a.p, drop=a.drop, peaky=a.peaky)}
    ms, qpool = timed(lambda: _capi.block_pool(q, nimg), a.iters)
    ms2, kpool = timed(lambda: _capi.block_pool(k, nb), a.iters)
    res["pool_ms"] = ms + ms2
    res["pool_GBps"] = (q.numel() * 2 * (nimg / nb) + k.numel() * 2) / ((ms + ms2) * 1e-3) / 1e9
    ms, (mask, idx, cnt) = timed(lambda: _capi.block_select(qpool, kpool, nbm, nimg, tb, top_k, a.p), a.iters)
    res["select_ms"] = ms
    mk = _capi.block_select(qpool, kpool, nbm, nimg, tb, top_k, a.p, want_mask=True, want_lists=False)[0].bool()
    both = (mk[:, :, :-1, :nimg] & mk[:, :, 1:, :nimg]).sum(-1).float()
    res["adjacent_shared_frac"] = float((both / mk[:, :, :-1, :nimg].sum(-1).float()).mean().item())
    del mk, both
    ms, vt = timed(lambda: _capi.pack_v(v, nb), a.iters)
    res["pack_v_ms"] = ms
    res["pack_v_GBps"] = 2 * v.numel() * 2 / (ms * 1e-3) / 1e9
    if a.same_list:
        n = int(cnt.float().mean().item())
        idx = torch.arange(nb, device=dev, dtype=torch.int32).expand(1, H, nimg, nb).contiguous()
        idx[..., n - tb:n] = torch.arange(nimg, nb, device=dev, dtype=torch.int32)
        cnt = torch.full_like(cnt, n)
    if a.l2_resident:
        n = int(cnt.float().mean().item())
        idx = (torch.arange(nb, device=dev, dtype=torch.int32) % a.l2_resident).expand(1, H, nimg, nb).contiguous()
        cnt = torch.full_like(cnt, n)
    if a.window:
        n = max(1, int(0.316 * a.window))
        r = torch.rand(1, H, nimg, a.window, device=dev)
        sel = r.topk(n, dim=-1).indices.sort(dim=-1).values.to(torch.int32)
        idx = torch.zeros(1, H, nimg, nb, dtype=torch.int32, device=dev)
        idx[..., :n] = sel
        cnt = torch.full_like(cnt, n)
    if a.pair_overlap >= 0:
        n = int(cnt.float().mean().item())
        g = torch.Generator(device=dev).manual_seed(3)
        r = torch.rand(1, H, nimg, nimg, device=dev, generator=g)
        # odd rows reuse the even row's random keys for a fraction of the columns -> correlated top-n sets
        take = torch.rand(1, H, nimg, nimg, device=dev, generator=g) < a.pair_overlap
        r[:, :, 1::2] = torch.where(take[:, :, 1::2], r[:, :, 0:nimg - 1:2][:, :, : r[:, :, 1::2].shape[2]], r[:, :, 1::2])
        sel = r.topk(n - tb, dim=-1).indices.sort(dim=-1).values.to(torch.int32)
        idx = torch.zeros(1, H, nimg, nb, dtype=torch.int32, device=dev)
        idx[..., : n - tb] = sel
        idx[..., n - tb:n] = torch.arange(nimg, nb, device=dev, dtype=torch.int32)
        cnt = torch.full_like(cnt, n)
        ev, od = idx[:, :, 0:nimg - 1:2, : n - tb], idx[:, :, 1::2, : n - tb]
        ev = ev[:, :, : od.shape[2]]
        m_ev = torch.zeros(1, H, od.shape[2], nb, dtype=torch.bool, device=dev).scatter_(-1, ev.long(), True)
        res["pair_shared_frac"] = float(m_ev.gather(-1, od.long()).float().mean().item())
    if a.k_head_major:
        k = k.permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3)
        q = q.permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3)
    kept = int(cnt.sum().item())
    pairs = kept + H * tb * nb
    flops = 4 * 128 ** 3 * pairs
    res["pairs"] = pairs
    res["kept_mean"] = kept / (H * nimg)
    res["kept_min_max"] = [int(cnt.min()), int(cnt.max())]
    fl = a.flags
    if a.sorted:
        fl = (_capi.ATTN_DEFAULT_FLAGS if fl is None else fl) | _capi.ATTN_SORTED
    res["flags"] = _capi.ATTN_DEFAULT_FLAGS if fl is None else fl
    ms, o = timed(lambda: _capi.bsattn_fwd(q, k, vt, seqlens, idx, cnt, nimg, 128 ** -0.5, 0.0, nimg,
                                           xcd_remap=not a.no_xcd, flags=fl), a.iters)
    res["attn_ms"] = ms
    res["attn_TFLOPs"] = flops / (ms * 1e-3) / 1e12
    res["attn_frac_of_2.5PF"] = res["attn_TFLOPs"] / 2500
    res["dense_equiv_TFLOPs"] = 4 * S * S * 128 * H / (ms * 1e-3) / 1e12
    res["finite"] = bool(torch.isfinite(o.float()).all().item())
    for fl2 in a.also_flags:
        ms2, o2 = timed(lambda: _capi.bsattn_fwd(q, k, vt, seqlens, idx, cnt, nimg, 128 ** -0.5, 0.0, nimg,
                                                 xcd_remap=not a.no_xcd, flags=fl2), a.iters)
        d = (o2.float() - o.float()).abs()
        res.setdefault("also", {})[str(fl2)] = dict(attn_ms=ms2, attn_TFLOPs=flops / (ms2 * 1e-3) / 1e12,
                                                    frac=flops / (ms2 * 1e-3) / 1e12 / 2500,
                                                    max_abs_diff_vs_first=float(d.max().item()),
                                                    mean_abs_diff_vs_first=float(d.mean().item()),
                                                    finite=bool(torch.isfinite(o2.float()).all().item()))
        del o2, d
    # second pass of the first set: the board reaches its power steady state over the run, order effects show here
    if a.also_flags:
        ms3, _ = timed(lambda: _capi.bsattn_fwd(q, k, vt, seqlens, idx, cnt, nimg, 128 ** -0.5, 0.0, nimg,
                                                xcd_remap=not a.no_xcd, flags=fl), a.iters)
        res["attn_ms_second_pass"] = ms3
    print(json.dumps(res))


if __name__ == "__main__":
    main()
