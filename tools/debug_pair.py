#!/usr/bin/env python3
"""Debug aid (GPU): the pair kernel against the round-1 kernel on small cases, with a map of the rows that differ."""
import itertools
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jenga_amd import _capi  # noqa: E402

dev = torch.device("cuda:0")


def lists_from_mask(m):
    order = torch.argsort((~m).to(torch.int8), dim=-1, stable=True).to(torch.int32)
    return order.contiguous(), m.sum(-1).to(torch.int32).contiguous()


def case(name, H, nq_img, tb, mode, valid_text, amp, scale=1.3, density=0.4, seed=1, dt=torch.bfloat16):
    g = torch.Generator().manual_seed(seed)
    nb = nq_img + tb
    S = nb * 128
    q = (torch.randn(1, S, H, 128, generator=g) * scale).to(dt).to(dev)
    k = (torch.randn(1, S, H, 128, generator=g) * scale).to(dt).to(dev)
    v = torch.randn(1, S, H, 128, generator=g).to(dt).to(dev)
    mask = torch.rand(1, H, nq_img, nb, generator=g) < density
    if mode == "shared":
        mask[:, :, 1::2] = mask[:, :, 0:nq_img - 1:2][:, :, :mask[:, :, 1::2].shape[2]]
    elif mode == "disjoint":
        mask[:, :, 1::2] = ~mask[:, :, 0:nq_img - 1:2][:, :, :mask[:, :, 1::2].shape[2]]
    mask[..., nq_img:] = True
    if mode != "disjoint":
        for m in range(nq_img):
            mask[:, :, m, m] = True
    mask = mask.to(dev)
    idx, cnt = lists_from_mask(mask)
    vt = _capi.pack_v(v, nb)
    seqlen = nq_img * 128 + valid_text if tb else nq_img * 128 - 19
    sl = torch.tensor([seqlen], dtype=torch.int32, device=dev)
    run = lambda fl: _capi.bsattn_fwd(q, k, vt, sl, idx, cnt, nq_img, 128 ** -0.5, amp, nq_img, flags=fl)
    ref = run(1).float()
    lp = run(9).float()
    pr = run(65).float()
    torch.cuda.synchronize()
    d = (pr - ref).abs()
    d[~torch.isfinite(d)] = 99.0
    dl = (lp - ref).abs()
    per_block = d.view(1, nb, 128, H, 128).amax(dim=(2, 4))[0]          # [nb, H]
    bad = (per_block > 0.03).nonzero().tolist()
    print(f"{name:28s} H{H} nq{nq_img} tb{tb} {mode:8s} vt{valid_text} amp{amp} sc{scale}: max|pair-ref|={d.max().item():.4f} "
          f"max|lp-ref|={dl.max().item():.4f} nonfinite={int((~torch.isfinite(pr)).sum())} bad (qblock, head)={bad[:12]}")
    if bad:
        pidx, pcnt = _capi.pair_merge(idx, cnt, nb)
        for qb, h in bad[:4]:
            if qb < nq_img:
                print("      pair", qb // 2, "sub", "AB"[qb & 1], "n_sh,n_a,n_b =", pcnt[0, h, qb // 2, :3].tolist(),
                      "rows bad:", (d.view(1, nb, 128, H, 128)[0, qb, :, h].amax(-1) > 0.03).nonzero().flatten().tolist()[:8])
            else:
                print("      text block", qb - nq_img)


if __name__ == "__main__" and "--text" not in sys.argv:
    case("seed1-like", 2, 8, 2, "random", 37, 0.431)
    case("amp0", 2, 8, 2, "random", 37, 0.0)
    case("full text", 2, 8, 2, "random", 256, 0.0)
    case("full text amp", 2, 8, 2, "random", 256, 0.431)
    case("shared only", 2, 8, 2, "shared", 256, 0.0)
    case("shared vt37", 2, 8, 2, "shared", 37, 0.0)
    case("disjoint", 2, 8, 2, "disjoint", 256, 0.0)
    case("no text", 2, 8, 0, "random", 0, 0.0)
    case("odd nq", 2, 9, 2, "random", 100, 0.0)
    case("one text block", 2, 6, 1, "random", 100, 0.0)
    case("big scale (exact path)", 2, 16, 2, "random", 256, 0.0, scale=3.0)
    case("big scale shared", 2, 16, 2, "shared", 256, 0.0, scale=3.0)
    case("big scale disjoint", 2, 16, 2, "disjoint", 256, 0.0, scale=3.0)
    case("longer lists", 2, 64, 2, "random", 64, 0.0, density=0.5)
    case("longer shared", 2, 64, 2, "shared", 64, 0.0, density=0.5)
    case("longer shared big", 2, 64, 2, "shared", 64, 0.0, density=0.5, scale=2.5)
    case("fp16", 2, 16, 2, "random", 64, 0.2, dt=torch.float16)


def text_detail():
    H, nq_img, tb = 1, 4, 2
    g = torch.Generator().manual_seed(1)
    nb = nq_img + tb
    S = nb * 128
    q = (torch.randn(1, S, H, 128, generator=g) * 1.0).to(torch.bfloat16).to(dev)
    k = (torch.randn(1, S, H, 128, generator=g) * 1.0).to(torch.bfloat16).to(dev)
    v = torch.randn(1, S, H, 128, generator=g).to(torch.bfloat16).to(dev)
    mask = torch.ones(1, H, nq_img, nb, dtype=torch.bool, device=dev)
    idx, cnt = lists_from_mask(mask)
    vt = _capi.pack_v(v, nb)
    sl = torch.tensor([S], dtype=torch.int32, device=dev)
    for vv, tag in ((v, "random V"), (torch.ones_like(v), "V = 1")):
        vt = _capi.pack_v(vv, nb)
        ref = _capi.bsattn_fwd(q, k, vt, sl, idx, cnt, nq_img, 128 ** -0.5, 0.0, nq_img, flags=1).float()
        pr = _capi.bsattn_fwd(q, k, vt, sl, idx, cnt, nq_img, 128 ** -0.5, 0.0, nq_img, flags=65).float()
        torch.cuda.synchronize()
        d = (pr - ref).abs()[0, :, 0]          # [S, 128]
        rows = d.amax(-1).view(nb, 128)
        print(tag, "per-block max err:", [round(float(x), 3) for x in rows.amax(-1)])
        tb1 = rows[nb - 1]
        print("  text block 1 rows err (first 40):", [round(float(x), 2) for x in tb1[:40]])
        cols = d.view(nb, 128, 128)[nb - 1].amax(0)
        print("  text block 1 cols err (first 40):", [round(float(x), 2) for x in cols[:40]])
        print("  pair out row0[:8]", [round(float(x), 3) for x in pr[0, (nb - 1) * 128, 0, :8]], "ref", [round(float(x), 3) for x in ref[0, (nb - 1) * 128, 0, :8]])
        print("  pair out row0 of text block 0[:8]", [round(float(x), 3) for x in pr[0, (nb - 2) * 128, 0, :8]], "ref", [round(float(x), 3) for x in ref[0, (nb - 2) * 128, 0, :8]])


if __name__ == "__main__" and "--text" in sys.argv:
    text_detail()
