import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from jenga_amd import _capi
dev = torch.device("cuda:0")
x = torch.randn(1, 75600, 5120, device=dev)
sh, sc = torch.randn(1, 5120, device=dev) * 0.3, torch.randn(1, 5120, device=dev) * 0.3
w, b = torch.randn(5120, device=dev), torch.randn(5120, device=dev)
y = torch.randn(1, 75600, 5120, device=dev).to(torch.bfloat16)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
ms = t(lambda: _capi.wan_ln_modulate(x, shift=sh, scale=sc, eps=1e-6)); print("ln_modulate(mod)  %.3f ms  %.2f TB/s" % (ms, 75600*5120*6/ms/1e9))
ms = t(lambda: _capi.wan_ln_modulate(x, weight=w, bias=b, eps=1e-6)); print("ln_modulate(affine) %.3f ms  %.2f TB/s" % (ms, 75600*5120*6/ms/1e9))
ms = t(lambda: _capi.wan_gate_residual(x, y, sc)); print("gate_residual      %.3f ms  %.2f TB/s" % (ms, 75600*5120*10/ms/1e9))
ms = t(lambda: _capi.wan_gate_residual(x, y, sc, out=x)); print("gate_residual inplace %.3f ms  %.2f TB/s" % (ms, 75600*5120*10/ms/1e9))
