#!/usr/bin/env python3
"""Host time per call of the ctypes wrappers (GPU box): how long Python + ctypes take to ENQUEUE one kernel, measured on
tensors so small that the GPU is never the limit.  The per-rank steps of the low-resolution stages of an 8-rank job issue
~1900 launches in ~100 ms: at 30-50 us of host time per launch they are host-bound."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jenga_amd import _capi  # noqa: E402


def per_call(fn, n=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6


def main():
    dev = torch.device("cuda:0")
    bf = torch.bfloat16
    x = torch.randn(1, 128, 3072, device=dev, dtype=bf)
    vec = torch.randn(1, 3072, device=dev, dtype=bf)
    w = torch.randn(3072, 3072, device=dev, dtype=bf) * 0.02
    b = torch.randn(3072, device=dev, dtype=bf)
    qkv = torch.randn(1, 128, 3, 24, 128, device=dev, dtype=bf)
    wn = torch.ones(128, device=dev, dtype=bf)
    q = torch.empty(1, 128, 24, 128, device=dev, dtype=bf)
    k = torch.empty_like(q)
    out = {
        "torch.nn.functional.linear": per_call(lambda: torch.nn.functional.linear(x, w, b)),
        "_capi.linear (bias)": per_call(lambda: _capi.linear(x, w, b)),
        "_capi.linear (gate + residual)": per_call(lambda: _capi.linear(x, w, b, gate=vec, res=x)),
        "_capi.ln_modulate": per_call(lambda: _capi.ln_modulate(x, vec, vec)),
        "_capi.qk_norm_rope_pool": per_call(lambda: _capi.qk_norm_rope_pool(qkv[:, :, 0], qkv[:, :, 1], wn, wn, None, None, q, k)),
        "_capi.pack_v": per_call(lambda: _capi.pack_v(qkv[:, :, 2], 1)),
        "torch.empty": per_call(lambda: torch.empty((1, 128, 3072), dtype=bf, device=dev)),
    }
    print(json.dumps({k_: round(v, 2) for k_, v in out.items()}))


if __name__ == "__main__":
    main()
