import torch, sys
sys.path.insert(0, '.')
from jenga_amd import _capi
dev = torch.device('cuda:0')
for dt in (torch.float16, torch.bfloat16):
    g = torch.Generator(device=dev).manual_seed(2405)
    H, nimg, ntxt = 24, 5, 2
    nb = nimg + ntxt; S, S_img = nb*128, nimg*128
    lin = (torch.randn(1, S, 3*H*128+64, generator=g, device=dev)*1.7).to(dt)
    qkv = lin[..., :3*H*128].unflatten(-1, (3, H, 128))
    xq, xk = qkv[:, :, 0], qkv[:, :, 1]
    wq = (1 + 0.1*torch.randn(128, generator=g, device=dev)).to(dt)
    cos = torch.randn(S_img, 128, generator=g, device=dev); sin = torch.randn(S_img, 128, generator=g, device=dev)
    for w in (wq, None):
        for cs in ((cos, sin), (None, None)):
            q_ref = _capi.rmsnorm_rope(xq, w, cs[0], cs[1], s_rope=S_img if cs[0] is not None else None)
            k_ref = _capi.rmsnorm_rope(xk, w, cs[0], cs[1], s_rope=S_img if cs[0] is not None else None)
            q = torch.empty_like(q_ref); k = torch.empty_like(q_ref)
            _capi.qk_norm_rope_pool(xq, xk, w, w, cs[0], cs[1], q, k, s_rope=S_img if cs[0] is not None else None)
            torch.cuda.synchronize()
            d = (q != q_ref)
            print(dt, 'w' if w is not None else '-', 'rope' if cs[0] is not None else '-', 'mismatch', int(d.sum()), 'of', d.numel(),
                  'max abs diff', float((q.float()-q_ref.float()).abs().max()))
            if d.any():
                i = d.nonzero()[0].tolist(); print('  first at', i, float(q[tuple(i)]), float(q_ref[tuple(i)]), 'x=', float(xq[tuple(i)]))

# ---- which kernel follows the arithmetic as written?  (fp16, no weight, no rope)
import numpy as np
dt = torch.float16
g = torch.Generator(device=dev).manual_seed(2405)
H, nimg, ntxt = 24, 5, 2
nb = nimg + ntxt; S, S_img = nb*128, nimg*128
lin = (torch.randn(1, S, 3*H*128+64, generator=g, device=dev)*1.7).to(dt)
qkv = lin[..., :3*H*128].unflatten(-1, (3, H, 128))
xq, xk = qkv[:, :, 0], qkv[:, :, 1]
q_ref = _capi.rmsnorm_rope(xq, None, None, None)
q = torch.empty_like(q_ref); k = torch.empty_like(q_ref)
_capi.qk_norm_rope_pool(xq, xk, None, None, None, None, q, k)
torch.cuda.synchronize()
d = (q != q_ref).nonzero()
for i in d[:3].tolist():
    b, s, h, e = i
    x = xq[b, s, h].float().cpu().numpy().astype(np.float32)
    part = np.zeros(16, np.float32)
    for sub in range(16):
        ss = np.float32(0)
        for j in range(8):
            ss = np.float32(ss + np.float32(x[sub*8+j] * x[sub*8+j]))
        part[sub] = ss
    for o in (1, 2, 4, 8):
        part = np.array([np.float32(part[l] + part[l ^ o]) for l in range(16)], np.float32)
    ss = part[e // 8]
    r = np.float32(1.0) / np.sqrt(np.float32(np.float32(ss / np.float32(128)) + np.float32(1e-6)), dtype=np.float32)
    y = np.float16(np.float32(x[e] * np.float32(r)))
    print(i, 'numpy', float(y), 'fused', float(q[b, s, h, e]), 'ref', float(q_ref[b, s, h, e]), 'r', float(r))
