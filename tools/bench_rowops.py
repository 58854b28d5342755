#!/usr/bin/env python3
"""Micro-benchmark of the fused Q / K RMSNorm + RoPE + pooling kernel at the HunyuanVideo 720p shape (the q / k slices of one QKV
GEMM output, 115 456 tokens x 24 heads): ms, GB/s against the algorithmic bytes, and a checksum of every output (A/B builds must
agree bit for bit).  JENGA_LIB selects the library build.  python tools/bench_rowops.py [--reps 30]"""
import argparse
import hashlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jenga_amd import _capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(1)
    H, nimg, ntxt = 24, 900, 2
    nb = nimg + ntxt
    S, S_img = nb * 128, nimg * 128
    lin = (torch.randn(1, S, 3 * H * 128, generator=g, device=dev) * 1.3).to(torch.bfloat16)
    qkv = lin.unflatten(-1, (3, H, 128))
    xq, xk = qkv[:, :, 0], qkv[:, :, 1]
    wq = (1 + 0.1 * torch.randn(128, generator=g, device=dev)).to(torch.bfloat16)
    wk = (1 + 0.1 * torch.randn(128, generator=g, device=dev)).to(torch.bfloat16)
    ang = torch.rand(S_img, 64, generator=g, device=dev) * 6.28
    cos = torch.cos(ang).repeat_interleave(2, dim=1).contiguous()
    sin = torch.sin(ang).repeat_interleave(2, dim=1).contiguous()
    oq = torch.empty(1, S, H, 128, dtype=torch.bfloat16, device=dev)
    ok = torch.empty_like(oq)
    qp = torch.zeros(1, H, nimg, 128, dtype=torch.bfloat16, device=dev)
    kp = torch.zeros(1, H, nb, 128, dtype=torch.bfloat16, device=dev)

    def run():
        _capi.qk_norm_rope_pool(xq, xk, wq, wk, cos, sin, oq, ok, s_rope=S_img, qpool=qp, kpool=kp)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    nbytes = 4 * S * H * 128 * 2 + 2 * S_img * 128 * 4 + (nimg + nb) * H * 128 * 2
    h = hashlib.sha256()
    for t in (oq, ok, qp, kp):
        h.update(t.cpu().view(torch.uint8).numpy().tobytes())
    print(json.dumps({"lib": os.environ.get("JENGA_LIB", "product"), "qk_norm_rope_pool_ms": round(ms, 4),
                      "GBps": round(nbytes / ms / 1e6, 1), "frac_of_8TBps": round(nbytes / ms / 1e6 / 8000, 4),
                      "outputs_sha256": h.hexdigest()[:16]}))


if __name__ == "__main__":
    main()
