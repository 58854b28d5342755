import torch, time, sys
sys.path.insert(0, '.')
from jenga_amd import _capi
dev = torch.device('cuda:0')
S, C, Hd = 115200, 3072, 12288
x = torch.randn(S, C, device=dev, dtype=torch.bfloat16)
w = torch.randn(Hd, C, device=dev, dtype=torch.bfloat16) * 0.02
b = torch.randn(Hd, device=dev, dtype=torch.bfloat16) * 0.02
def t(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): y = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, y
def sep():
    h = torch.nn.functional.linear(x, w, b)
    return _capi.gelu_tanh(h.view(1, S, Hd), out=h.view(1, S, Hd))
def epi():
    return torch._addmm_activation(b, x, w.t(), use_gelu=True)
ms1, y1 = t(sep); ms2, y2 = t(epi)
print("separate %.2f ms, epilogue %.2f ms, max diff %.4f, mean diff %.5f" % (ms1, ms2, (y1.float().view(S,Hd)-y2.float()).abs().max().item(), (y1.float().view(S,Hd)-y2.float()).abs().mean().item()))
ms3, _ = t(lambda: torch.nn.functional.linear(x, w, b)); print("linear alone %.2f ms" % ms3)
