"""Diagnostics: ONE process, two streams -- a load (torch GEMM + elementwise) on one stream, a kernel repeated on fixed inputs
on another: does the result change?  (tools/diag_victim.py is the two-process form.)  JENGA_LIB selects the library build."""
import json, os, sys, threading, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jenga_amd import _capi

dev = torch.device("cuda", 0)
secs = float(os.environ.get("DIAG_SECS", "10"))
g = torch.Generator(device=dev).manual_seed(7)
C = 1024
xx = torch.randn(1, 512, C, generator=g, device=dev, dtype=torch.bfloat16)
sh_ = torch.randn(1, C, generator=g, device=dev, dtype=torch.bfloat16)
sc_ = torch.randn(1, C, generator=g, device=dev, dtype=torch.bfloat16)
a = torch.randn(2048, 2048, generator=g, device=dev, dtype=torch.bfloat16)
stop = False
n_load = [0]


def load():
    s = torch.cuda.Stream()
    if mode.startswith("spin"):       # MFMA-only kernels of tools/micro/mfma_spin.hip: spin32 / spin16 / spinvalu [blocks]
        import ctypes
        lib = ctypes.CDLL(os.path.join(ROOT, "tools", "micro", "bin", "libmfma_spin.so"))
        sink = torch.zeros(4, device=dev)
        kind = {"spin32": 0, "spin16": 1, "spinvalu": 2}[mode]
        blocks = int(os.environ.get("DIAG_SPIN_BLOCKS", "2048"))
        while not stop:
            lib.mfma_spin(ctypes.c_void_p(s.cuda_stream), ctypes.c_void_p(sink.data_ptr()), kind, blocks, 4000)
            n_load[0] += 1
            if n_load[0] % 8 == 0:
                s.synchronize()
        s.synchronize()
        return
    with torch.cuda.stream(s):
        b = a.clone()
        while not stop:
            if mode in ("torch", "mfma"):
                b = a @ a                       # MFMA GEMM (hipBLASLt)
            if mode in ("torch", "valu"):
                b = torch.nn.functional.layer_norm(b.tanh(), (2048,))      # VALU / memory only
            if mode == "copy":
                b.copy_(a)
            n_load[0] += 1
            if n_load[0] % 64 == 0:
                s.synchronize()
    s.synchronize()


mode = os.environ.get("DIAG_LOAD", "torch")
t = threading.Thread(target=load) if mode != "none" else None
if t:
    t.start()
first = _capi.ln_modulate(xx, sh_, sc_)
torch.cuda.synchronize()
bad = runs = 0
t0 = time.time()
while time.time() - t0 < secs:
    o = _capi.ln_modulate(xx, sh_, sc_)
    if not torch.equal(o, first):
        bad += 1
    runs += 1
stop = True
if t:
    t.join()
print(json.dumps({"lib": os.environ.get("JENGA_LIB", "product"), "load": mode, "runs": runs, "ln_modulate_mismatches": bad,
                  "load_iterations": n_load[0]}))
