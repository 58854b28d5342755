#!/usr/bin/env python3
"""Experiment (round 4, K1): cross-XCD balancing of the attention launch (JENGA_ATTN_BALANCE) against the static mapping,
same box, interleaved, at the HunyuanVideo 720p shape (flat lists).  Also dumps the per-workgroup [start, end] ticks of one
balanced launch (JENGA_LP_TIMES_DUMP) so that the per-XCD finish times can be read.  One JSON line.
  python tools/balance_ab.py [--iters 40] [--dump DIR]      (--dump: the tick dump is in the experiments library only,
  JENGA_LIB=$PWD/jenga_amd/libjenga_amd_exp.so, and records the rotated variants)"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jenga_amd import _capi, gilbert as G  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--heads", type=int, default=24)
    ap.add_argument("--dump", default="")
    ap.add_argument("--coherent", action="store_true", help="also a case with clustered (peaky) lists")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    t, h, w = 32, 45, 80
    S_img, tb = t * h * w, 2
    nimg, nb = S_img // 128, S_img // 128 + tb
    S, H = nb * 128, a.heads
    nbm = G.gilbert_block_neighbor_mapping(t, h, w, as_tensor=True)
    seqlens = torch.tensor([S_img + 64], dtype=torch.int32, device=dev)

    def case(seed, drop, peaky=0.0):
        g = torch.Generator(device=dev).manual_seed(seed)
        q, k, v = (torch.randn(1, S, H, 128, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(3))
        if peaky > 0:
            cent = torch.randn(1, nb, 1, H, 128, device=dev, generator=g) * peaky
            k = (k.view(1, nb, 128, H, 128).float() + cent).to(torch.bfloat16).view(1, S, H, 128)
            pick = torch.randint(0, nimg, (nb,), device=dev, generator=g)
            q = (q.view(1, nb, 128, H, 128).float() + cent[:, pick]).to(torch.bfloat16).view(1, S, H, 128)
        qp, kp = _capi.block_pool(q, nimg), _capi.block_pool(k, nb)
        _, idx, cnt = _capi.block_select(qp, kp, nbm, nimg, tb, int((1 - drop) * nimg), 0.3)
        vt = _capi.pack_v(v, nb)
        pairs = int(cnt.sum().item()) + H * tb * nb
        return dict(q=q, k=k, vt=vt, idx=idx, cnt=cnt, pairs=pairs,
                    kept=[int(cnt.min()), round(float(cnt.float().mean()), 1), int(cnt.max())])

    def run(c, flags, iters=None, warm=3, env=None):
        iters = iters or a.iters
        for k_, v_ in (env or {}).items():
            os.environ[k_] = v_
        fn = lambda: _capi.bsattn_fwd(c["q"], c["k"], c["vt"], seqlens, c["idx"], c["cnt"], nimg, 128 ** -0.5, 0.0, nimg,
                                      flags=flags)
        for _ in range(warm):
            o = fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            o = fn()
        e1.record()
        torch.cuda.synchronize()
        for k_ in (env or {}):
            os.environ.pop(k_)
        ms = e0.elapsed_time(e1) / iters
        return round(4 * 128 ** 3 * c["pairs"] / (ms * 1e-3) / 1e12, 1), o

    base = _capi.ATTN_DEFAULT_FLAGS
    bal = base | _capi.ATTN_BALANCE
    rot = base | _capi.ATTN_ROTATE
    both = bal | _capi.ATTN_ROTATE
    res = {"iters": a.iters, "unit": "TFLOP/s"}
    cases = [("flat70", case(0, 0.7))]
    for name, c in cases:
        r = {"kept_min_mean_max": c["kept"]}
        r["default"], o0 = run(c, base)
        r["balance"], o1 = run(c, bal)
        r["bit_identical"] = bool(torch.equal(o0, o1))
        del o0, o1
        r["default_2"], _ = run(c, base)
        r["balance_2"], _ = run(c, bal)
        for pct in ("0", "6", "25"):
            r["balance_extra_" + pct], _ = run(c, bal, env={"JENGA_BALANCE_EXTRA_PCT": pct})
        r["rotate_clock"], _ = run(c, rot)
        r["balance_rotate"], _ = run(c, both)
        r["rotate_clock_2"], _ = run(c, rot)
        r["balance_rotate_2"], _ = run(c, both)
        r["default_3"], _ = run(c, base)
        r["balance_3"], _ = run(c, bal)
        if a.dump:
            os.makedirs(a.dump, exist_ok=True)
            run(c, bal, iters=3, warm=0, env={"JENGA_LP_TIMES_DUMP": os.path.join(a.dump, name + "_balance.times")})
            run(c, rot, iters=3, warm=0, env={"JENGA_LP_TIMES_DUMP": os.path.join(a.dump, name + "_rotate.times")})
        res[name] = r
    if a.coherent:
        del cases
        c = case(0, 0.85, peaky=1.0)
        r = {"kept_min_mean_max": c["kept"]}
        r["default"], _ = run(c, base)
        r["balance"], _ = run(c, bal)
        r["default_2"], _ = run(c, base)
        r["balance_2"], _ = run(c, bal)
        res["peaky85"] = r
    print(json.dumps(res))


if __name__ == "__main__":
    main()
