#!/usr/bin/env python3
"""The DiT's GEMM shapes in a sustained loop (for tools/power_sample.py): prints TFLOP/s per shape over ~SECONDS each.
  python tools/power_sample.py --out gpurun_out/x.json -- python tools/gemm_loop.py [--seconds 6]"""
import argparse
import json
import time

import torch

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=6.0)
a = ap.parse_args()
dev = torch.device("cuda:0")
S = 115200
shapes = {"qkv 3072->9216": (3072, 9216), "proj 3072->3072": (3072, 3072), "fc1 3072->12288": (3072, 12288),
          "fc2 12288->3072": (12288, 3072), "linear1 3072->21504": (3072, 21504), "linear2 15360->3072": (15360, 3072)}
res = {}
for name, (k, n) in shapes.items():
    x = torch.randn(S, k, device=dev, dtype=torch.bfloat16)
    w = torch.randn(n, k, device=dev, dtype=torch.bfloat16) * 0.02
    b = torch.zeros(n, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        torch.addmm(b, x, w.t())
    torch.cuda.synchronize()
    t0 = time.time()
    it = 0
    while time.time() - t0 < a.seconds:
        for _ in range(20):
            torch.addmm(b, x, w.t())
        torch.cuda.synchronize()
        it += 20
    dt = time.time() - t0
    res[name] = round(2.0 * S * k * n * it / dt / 1e12, 1)
    del x, w
print(json.dumps({"gemm_TFLOPs": res, "M": S}))
