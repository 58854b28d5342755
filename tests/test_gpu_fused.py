"""GPU (-m gpu): the fused Q+K RMSNorm + RoPE + block-pooling kernel against the kernels it
replaces -- which are themselves pinned to the reference goldens in test_gpu_parity.py -- bit for bit."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("H,nimg,ntxt", [(24, 5, 2), (3, 2, 0), (40, 3, 1)])
def test_qk_norm_rope_pool_equals_separate_kernels(dev, dt, H, nimg, ntxt):
    from jenga_amd import _capi
    g = torch.Generator(device=dev).manual_seed(H * 100 + nimg)
    nb = nimg + ntxt
    S, S_img = nb * 128, nimg * 128
    B = 1
    # q and k as slices of one fused QKV GEMM output (strided), plus some extra trailing columns like linear1's MLP part
    lin = (torch.randn(B, S, 3 * H * 128 + 64, generator=g, device=dev) * 1.7).to(dt)
    qkv = lin[..., : 3 * H * 128].unflatten(-1, (3, H, 128))
    xq, xk = qkv[:, :, 0], qkv[:, :, 1]
    wq = (1 + 0.1 * torch.randn(128, generator=g, device=dev)).to(dt)
    wk = (1 + 0.1 * torch.randn(128, generator=g, device=dev)).to(dt)
    cos = torch.randn(S_img, 128, generator=g, device=dev)
    sin = torch.randn(S_img, 128, generator=g, device=dev)
    # the replaced path
    q_ref = _capi.rmsnorm_rope(xq, wq, cos, sin, s_rope=S_img)
    k_ref = _capi.rmsnorm_rope(xk, wk, cos, sin, s_rope=S_img)
    qp_ref = _capi.block_pool(q_ref, nimg)
    kp_ref = _capi.block_pool(k_ref, nb)
    # fused, outputs written into views of larger buffers (what the double-stream blocks do)
    qbuf = torch.zeros(B, S + 128, H, 128, dtype=dt, device=dev)
    kbuf = torch.zeros_like(qbuf)
    qp = torch.zeros(B, H, nimg, 128, dtype=dt, device=dev)
    kp = torch.zeros(B, H, nb, 128, dtype=dt, device=dev)
    _capi.qk_norm_rope_pool(xq, xk, wq, wk, cos, sin, qbuf[:, :S], kbuf[:, :S], s_rope=S_img, qpool=qp, kpool=kp)
    torch.cuda.synchronize()
    assert torch.equal(qbuf[:, :S], q_ref) and torch.equal(kbuf[:, :S], k_ref)
    assert not qbuf[:, S:].any() and not kbuf[:, S:].any()
    assert torch.equal(qp, qp_ref) and torch.equal(kp, kp_ref)
    # two calls filling one pooled tensor: image stream (RoPE) then text stream (no RoPE, block offset), no weights
    if ntxt:
        qp2 = torch.zeros_like(qp)
        kp2 = torch.zeros_like(kp)
        q2 = torch.empty(B, S, H, 128, dtype=dt, device=dev)
        k2 = torch.empty_like(q2)
        _capi.qk_norm_rope_pool(xq[:, :S_img], xk[:, :S_img], None, None, cos, sin, q2[:, :S_img], k2[:, :S_img],
                                qpool=qp2, kpool=kp2)
        _capi.qk_norm_rope_pool(xq[:, S_img:], xk[:, S_img:], None, None, None, None, q2[:, S_img:], k2[:, S_img:],
                                qpool=None, kpool=kp2, pool_block0=nimg)
        q_ref2 = _capi.rmsnorm_rope(xq, None, cos, sin, s_rope=S_img)
        k_ref2 = _capi.rmsnorm_rope(xk, None, cos, sin, s_rope=S_img)
        torch.cuda.synchronize()
        assert torch.equal(q2, q_ref2) and torch.equal(k2, k_ref2)
        assert torch.equal(qp2, _capi.block_pool(q_ref2, nimg)) and torch.equal(kp2, _capi.block_pool(k_ref2, nb))


def test_qk_norm_rope_pool_rejects_bad_arguments(dev):
    from jenga_amd import _capi
    x = torch.zeros(1, 256, 2, 128, dtype=torch.bfloat16, device=dev)
    o = torch.empty_like(x)
    with pytest.raises(ValueError):
        _capi.qk_norm_rope_pool(x[:, :200], x[:, :200], None, None, None, None, o[:, :200], o[:, :200])   # S % 128
    with pytest.raises(ValueError):
        _capi.qk_norm_rope_pool(x, x.transpose(1, 2).contiguous().transpose(1, 2), None, None, None, None, o, o)
    with pytest.raises(ValueError):
        _capi.qk_norm_rope_pool(x, x, None, None, None, None, o, o, qpool=torch.zeros(1, 2, 2, 64, device=dev))
