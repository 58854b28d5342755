"""Independent victim test for the packed-fp32 finding (round 6, review item 4).

The round-5 diagnosis: `v_pk_{add,mul,fma}_f32 ... op_sel:[0,1]` returns wrong low-lane values while another wave of the GPU
executes `v_mfma_f32_16x16x32_bf16`.  All the evidence so far came from this library's own kernels and one stand-alone
reproducer.  This tool asks the question of code nobody here wrote or compiled: torch's own gfx950 kernels.

  * `tools/torch_pk_census.sh` (CPU, no GPU needed) disassembles every gfx950 code object of libtorch_hip.so and classifies the
    operand selection of every v_pk_{add,mul,fma}_f32: 1895 kernels use a non-default selection, 106 of them the BROADCAST-HIGH
    one (profiles/r06_torch_pk_selections.json.gz; the stand-alone reproducer's failing selection), the others SWAP /
    broadcast-low only (the census is profiles/r06_torch_pk_selections.json.gz).
  * This script repeats torch ops on fixed inputs on one stream -- ops whose kernels are on that list ("suspect") and ops
    whose kernels are not ("control") -- while a load runs on a second stream of the same process: nothing, 32x32x16 MFMAs,
    16x16x32 MFMAs (tools/micro/mfma_spin.hip: registers only, no memory traffic) or torch's bf16 GEMM (hipBLASLt, MI16x16).
    Every launch is a pure function of its inputs: any result that differs from the first one is a wrong result.
  * Under `rocprofv3 --kernel-trace --stats` the same command gives the kernel names each op really launched, to be intersected
    with the census (done by tools/torch_pk_census.sh --intersect).

Prints one JSON line per load.  DIAG_SECS seconds per load (default 8)."""
import ctypes
import json
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device("cuda", 0)
secs = float(os.environ.get("DIAG_SECS", "8"))
g = torch.Generator(device=dev).manual_seed(11)

N = 1 << 20
a32 = torch.rand(1024, 2048, generator=g, device=dev) + 0.5
b32 = torch.rand(1024, 2048, generator=g, device=dev) + 0.5
a16 = a32.to(torch.bfloat16)
b16_ = b32.to(torch.bfloat16)
ah = a32.half()
gemm_a = torch.randn(4096, 4096, generator=g, device=dev, dtype=torch.bfloat16)

c64a = torch.complex(a32[:, :1024].contiguous(), b32[:, :1024].contiguous())
c64b = torch.complex(b32[:, 1024:].contiguous(), a32[:, 1024:].contiguous())
va, vb = c64a[0].contiguous(), c64b[1].contiguous()

# (name, class, fn).  class "bcast": torch kernels that contain the BROADCAST-HIGH selection (op_sel:[0,1] / [1,0] with op_sel_hi
# left at 1: both result lanes read the high half of a source pair -- the selection of the stand-alone reproducer's failing forms;
# 106 kernels, almost all complex<float> arithmetic); True: kernels with the SWAP selection only (op_sel:[0,1] op_sel_hi:[1,0]);
# False: controls.  The trace intersect confirms per run which kernels were really launched.
OPS = [
    ("mul complex64 (MulFunctor<complex<float>>)", "bcast", lambda: c64a * c64b),
    ("mul complex64 strided", "bcast", lambda: c64a[:, ::2] * c64b[:, ::2]),
    ("add complex64 alpha=(0.5+0.25j) (CUDAFunctor_add<complex<float>>)", "bcast", lambda: torch.add(c64a, c64b, alpha=0.5 + 0.25j)),
    ("pow(complex64, complex64)", "bcast", lambda: torch.pow(c64a, c64b)),
    ("pow(complex64, 2.5)", "bcast", lambda: torch.pow(c64a, 2.5)),
    ("addr complex64", "bcast", lambda: torch.addr(c64a[:1024, :1024], va, vb)),
    ("prod complex64 dim=-1 (ReductionMulOp)", "bcast", lambda: torch.prod(c64a * 0.7, dim=-1)),
    ("cumprod complex64 (lookback_scan multiplies<complex>)", "bcast", lambda: torch.cumprod(c64a * 0.7, dim=-1)),
    ("_foreach_log complex64", "bcast", lambda: torch._foreach_log([c64a, c64b])[1]),
    ("polar", "bcast", lambda: torch.polar(a32, b32)),
    ("addcmul(fp32, value=tensor-like scalar) mixed dtypes", "bcast", lambda: torch.addcmul(a32, torch.tensor(1.5, device=dev), b16_)),
    ("mul fp32, strided operands (elementwise_kernel_manual_unroll MulFunctor<float>)", True,
     lambda: a32[:, ::2] * b32[:, ::2]),
    ("vector_norm fp32 dim=-1 (reduce_kernel NormOps<float>)", True, lambda: torch.linalg.vector_norm(a32, dim=-1)),
    ("vector_norm bf16 dim=-1 (reduce_kernel NormOps<BFloat16>)", True, lambda: torch.linalg.vector_norm(a16, dim=-1)),
    ("vector_norm fp32 full (reduce_kernel NormOps<float>)", True, lambda: torch.linalg.vector_norm(a32)),
    ("pow(tensor, tensor) fp32", True, lambda: torch.pow(a32, b32)),
    ("pow(tensor, tensor) fp32 strided", True, lambda: torch.pow(a32[:, ::2], b32[:, ::2])),
    ("pow(half, float)", True, lambda: torch.pow(ah, b32)),
    ("logaddexp fp32", True, lambda: torch.logaddexp(a32, b32)),
    ("logaddexp2 fp32", True, lambda: torch.logaddexp2(a32, b32)),
    ("softplus fp32", True, lambda: torch.nn.functional.softplus(a32)),
    ("mish fp32", True, lambda: torch.nn.functional.mish(a32)),
    ("log1p fp32", True, lambda: torch.log1p(a32)),
    ("asinh fp32", True, lambda: torch.asinh(a32)),
    ("acosh fp32", True, lambda: torch.acosh(a32 + 1.0)),
    ("logcumsumexp fp32", True, lambda: torch.logcumsumexp(a32, dim=-1)),
    ("renorm fp32", True, lambda: torch.renorm(a32, 2, 0, 1.0)),
    ("cdist fp32", True, lambda: torch.cdist(a32[:256], b32[:256])),
    # controls: kernels of the product loop, not on the census list
    ("add bf16 (control)", False, lambda: a16 + a16),
    ("bf16 -> fp32 copy (control)", False, lambda: a16.float()),
    ("layer_norm bf16 (control)", False, lambda: torch.nn.functional.layer_norm(a16, (2048,))),
    ("mul fp32 contiguous (control)", False, lambda: a32 * b32),
]

stop = False
n_load = [0]


def load(mode):
    s = torch.cuda.Stream()
    if mode.startswith("spin"):
        lib = ctypes.CDLL(os.path.join(ROOT, "tools", "micro", "bin", "libmfma_spin.so"))
        sink = torch.zeros(4, device=dev)
        kind = {"spin32": 0, "spin16": 1}[mode]
        while not stop:
            lib.mfma_spin(ctypes.c_void_p(s.cuda_stream), ctypes.c_void_p(sink.data_ptr()), kind, 2048, 4000)
            n_load[0] += 1
            if n_load[0] % 8 == 0:
                s.synchronize()
        s.synchronize()
        return
    with torch.cuda.stream(s):
        while not stop:
            _ = gemm_a @ gemm_a          # hipBLASLt bf16 GEMM (16x16x32 MFMAs)
            n_load[0] += 1
            if n_load[0] % 32 == 0:
                s.synchronize()
    s.synchronize()


def main():
    global stop
    modes = os.environ.get("DIAG_LOADS", "none,spin32,spin16,gemm").split(",")
    global OPS
    first, kept = {}, []
    for nm, cls_, f in OPS:          # an op this torch build rejects (dtype mix ...) is skipped, not fatal
        try:
            first[nm] = f().clone()
            kept.append((nm, cls_, f))
        except Exception as e:       # noqa: BLE001
            print(json.dumps({"skipped": nm, "why": repr(e)[:200]}), flush=True)
    OPS = kept
    torch.cuda.synchronize()
    for mode in modes:
        stop = False
        n_load[0] = 0
        t = threading.Thread(target=load, args=(mode,)) if mode != "none" else None
        if t:
            t.start()
            time.sleep(0.3)
        bad = {nm: 0 for nm, _, _ in OPS}
        runs = 0
        t0 = time.time()
        while time.time() - t0 < secs:
            for nm, _, f in OPS:
                if not torch.equal(f(), first[nm]):
                    bad[nm] += 1
            runs += 1
        stop = True
        if t:
            t.join()
        print(json.dumps({"load": mode, "load_iterations": n_load[0], "runs_per_op": runs,
                          "mismatches_broadcast_high_kernels": {nm: bad[nm] for nm, s_, _ in OPS if s_ == "bcast"},
                          "mismatches_swap_only_kernels": {nm: bad[nm] for nm, s_, _ in OPS if s_ is True},
                          "mismatches_control": {nm: bad[nm] for nm, s_, _ in OPS if s_ is False}}), flush=True)


if __name__ == "__main__":
    main()
