import torch, sys, time
sys.path.insert(0, '.')
from jenga_amd import _capi
dev = torch.device('cuda:0')
H, nimg, ntxt = 24, 900, 2
nb = nimg + ntxt; S, S_img = nb*128, nimg*128
lin = torch.randn(1, S, 3*H*128 + 12288, device=dev, dtype=torch.bfloat16)
qkv = lin[..., :3*H*128].unflatten(-1, (3, H, 128))
xq, xk = qkv[:, :, 0], qkv[:, :, 1]
w = torch.ones(128, device=dev, dtype=torch.bfloat16)
cos = torch.randn(S_img, 128, device=dev); sin = torch.randn(S_img, 128, device=dev)
q = torch.empty(1, S, H, 128, device=dev, dtype=torch.bfloat16); k = torch.empty_like(q)
qp = torch.empty(1, H, nimg, 128, device=dev, dtype=torch.bfloat16); kp = torch.empty(1, H, nb, 128, device=dev, dtype=torch.bfloat16)
def t(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
def sep():
    _capi.rmsnorm_rope(xq, w, cos, sin, s_rope=S_img, out=q); _capi.rmsnorm_rope(xk, w, cos, sin, s_rope=S_img, out=k)
    _capi.block_pool(q, nimg); _capi.block_pool(k, nb)
def fused():
    _capi.qk_norm_rope_pool(xq, xk, w, w, cos, sin, q, k, s_rope=S_img, qpool=qp, kpool=kp)
gb = (2*S*H*128*2*2 + 2*S_img*128*4)/1e9
a, b = t(sep), t(fused)
print("separate %.3f ms, fused %.3f ms (%.2f TB/s algorithmic)" % (a, b, gb/b))
