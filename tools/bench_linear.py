#!/usr/bin/env python3
"""The DiT's epilogue GEMMs through jenga_linear (hipBLASLt): ms per call for the library's first pick and, with
JENGA_GEMM_CANDIDATES=k in the environment, for the fastest of its first k picks.
  python tools/bench_linear.py [--ranks 8]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jenga_amd import _capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ranks", type=int, default=1)
a = ap.parse_args()
dev = torch.device("cuda:0")
S_img, S = 115200 // a.ranks, 115200 // a.ranks + 256
C, MLP = 3072, 12288
shapes = [  # name, M, K, N, kind
    ("double.proj+gate+res", S_img, C, C, "gate"),
    ("double.fc2+gate+res", S_img, MLP, C, "gate"),
    ("single.linear1.mlp+gelu (strided out)", S, C, MLP, "gelu"),
    ("single.linear2+gate+res", S, C + MLP, C, "gate"),
]
res = {}
g = torch.Generator(device=dev).manual_seed(0)
for name, M, K, N, kind in shapes:
    x = torch.randn(1, M, K, generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.bfloat16)
    b = torch.zeros(N, device=dev, dtype=torch.bfloat16)
    if kind == "gelu":
        cat = torch.empty(1, M, C + N, device=dev, dtype=torch.bfloat16)
        fn = lambda: _capi.linear(x, w, b, act=_capi.ACT_GELU_TANH, out=cat[..., C:])
    else:
        gate = torch.randn(1, N, generator=g, device=dev).to(torch.bfloat16)
        r = torch.randn(1, M, N, generator=g, device=dev).to(torch.bfloat16)
        o = torch.empty_like(r)
        fn = lambda: _capi.linear(x, w, b, gate=gate, res=r, out=o)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    res[name] = {"ms": round(ms, 4), "TFLOPs": round(2 * M * K * N / ms / 1e9, 1)}
print(json.dumps({"ranks": a.ranks, "candidates": os.environ.get("JENGA_GEMM_CANDIDATES", "1"), "gemms": res}))
