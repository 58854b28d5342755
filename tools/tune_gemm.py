"""Time the DiT's large GEMM shapes with the default hipBLASLt pick, and (with PYTORCH_TUNABLEOP_ENABLED=1 in the
environment) with TunableOp's pick; prints ms and TFLOP/s per shape.  Results file: $PYTORCH_TUNABLEOP_FILENAME."""
import sys
import time
import torch

dev = torch.device("cuda:0")
SHAPES = [  # (M, K, N, name)
    (115200, 3072, 9216, "double.img_qkv"),
    (115200, 3072, 3072, "double.img_proj"),
    (115200, 3072, 12288, "double.img_fc1"),
    (115200, 12288, 3072, "double.img_fc2"),
    (115456, 3072, 21504, "single.linear1"),
    (115456, 15360, 3072, "single.linear2"),
]
if len(sys.argv) > 1 and sys.argv[1].startswith("--ranks="):      # per-rank token counts of an N-rank Ulysses job
    n = int(sys.argv[1].split("=")[1])
    SHAPES = [(115200 // n + (256 if name.startswith("single") else 0), K, N, f"{name}/N{n}") for M, K, N, name in SHAPES]
elif len(sys.argv) > 1:
    SHAPES = [s for s in SHAPES if s[3] in sys.argv[1:]]
tot = 0.0
for M, K, N, name in SHAPES:
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
    b = torch.randn(N, device=dev, dtype=torch.bfloat16)
    t0 = time.time()
    y = torch.nn.functional.linear(x, w, b)
    torch.cuda.synchronize()
    first = time.time() - t0
    for _ in range(2):
        torch.nn.functional.linear(x, w, b)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        torch.nn.functional.linear(x, w, b)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    tot += ms
    print(f"{name:18s} M={M} K={K} N={N}  {ms:7.3f} ms  {2 * M * K * N / ms / 1e9:7.1f} TFLOP/s  (first call {first:.1f} s)", flush=True)
    del x, w, b, y
print(f"sum {tot:.2f} ms")
