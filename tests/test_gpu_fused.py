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


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,H,S_loc,S_txt", [(8, 24, 400, 256), (2, 8, 256, 256), (4, 8, 72, 512)])
def test_sp_qkv_prologue_equals_norm_rope_plus_pack(dev, dt, N, H, S_loc, S_txt):
    """jenga_sp_qkv_prologue (one launch: RMSNorm + RoPE of Q, K and the peer-major pack of Q, K, V; any shard length)
    against the five launches it replaces -- jenga_rmsnorm_rope x 2 + jenga_ulysses_pack_heads x 3 -- bit for bit, and
    the head-window form (a rank's own slice of the replicated text rows, written in place) against norm + slice."""
    from jenga_amd import _capi
    g = torch.Generator(device=dev).manual_seed(N * 1000 + S_loc)
    Hn = H // N
    S = S_loc + S_txt
    lin = (torch.randn(1, S, 3 * H * 128 + 64, generator=g, device=dev) * 1.7).to(dt)     # linear1's layout: qkv | mlp
    qkv = lin[..., : 3 * H * 128].unflatten(-1, (3, H, 128))
    wq = (1 + 0.1 * torch.randn(128, generator=g, device=dev)).to(dt)
    wk = (1 + 0.1 * torch.randn(128, generator=g, device=dev)).to(dt)
    cos = torch.randn(S_loc, 128, generator=g, device=dev)
    sin = torch.randn(S_loc, 128, generator=g, device=dev)
    img = [qkv[:, :S_loc, i] for i in range(3)]
    txt = [qkv[:, S_loc:, i] for i in range(3)]
    # ---- image rows -> peer-major send buffers
    sends = [torch.full((N, 1, S_loc, Hn, 128), 7.0, dtype=dt, device=dev) for _ in range(3)]
    _capi.sp_qkv_prologue(img[0], img[1], img[2], wq, wk, cos, sin, sends[0], sends[1], sends[2], Hn, s_rope=S_loc)
    q_ref = _capi.ulysses_pack_heads(_capi.rmsnorm_rope(img[0], wq, cos, sin, s_rope=S_loc), N)
    k_ref = _capi.ulysses_pack_heads(_capi.rmsnorm_rope(img[1], wk, cos, sin, s_rope=S_loc), N)
    v_ref = _capi.ulysses_pack_heads(img[2], N)
    torch.cuda.synchronize()
    assert torch.equal(sends[0], q_ref) and torch.equal(sends[1], k_ref) and torch.equal(sends[2], v_ref)
    # ---- text rows: rank r's head slice straight into the tail of the gathered attention inputs
    S_img = S_loc * N
    for r in (0, N - 1):
        fulls = [torch.zeros(1, S_img + S_txt, Hn, 128, dtype=dt, device=dev) for _ in range(3)]
        _capi.sp_qkv_prologue(txt[0], txt[1], txt[2], wq, wk, None, None, fulls[0][:, S_img:], fulls[1][:, S_img:],
                              fulls[2][:, S_img:], Hn, head0=r * Hn, n_heads=Hn)
        hs = slice(r * Hn, (r + 1) * Hn)
        tq = _capi.rmsnorm_rope(txt[0], wq, None, None)[:, :, hs]
        tk = _capi.rmsnorm_rope(txt[1], wk, None, None)[:, :, hs]
        torch.cuda.synchronize()
        assert torch.equal(fulls[0][:, S_img:], tq) and torch.equal(fulls[1][:, S_img:], tk)
        assert torch.equal(fulls[2][:, S_img:], txt[2][:, :, hs])
        assert not any(f[:, :S_img].any() for f in fulls)


def test_sp_qkv_prologue_rejects_bad_arguments(dev):
    from jenga_amd import _capi
    x = torch.zeros(1, 40, 8, 128, dtype=torch.bfloat16, device=dev)
    o = [torch.empty(4, 1, 40, 2, 128, dtype=torch.bfloat16, device=dev) for _ in range(3)]
    with pytest.raises(ValueError):      # q / k / v strides differ
        _capi.sp_qkv_prologue(x, x, x.transpose(1, 2).contiguous().transpose(1, 2), None, None, None, None, *o, 2)
    with pytest.raises(ValueError):      # N * Hn != H
        _capi.sp_qkv_prologue(x, x, x, None, None, None, None, *o, 4)
    with pytest.raises(ValueError):      # head window not on a peer boundary
        w = [torch.empty(1, 40, 2, 128, dtype=torch.bfloat16, device=dev) for _ in range(3)]
        _capi.sp_qkv_prologue(x, x, x, None, None, None, None, *w, 2, head0=1, n_heads=2)
    with pytest.raises(_capi.JengaError):   # the C ABI's own check (head window outside H)
        w = [torch.empty(1, 40, 2, 128, dtype=torch.bfloat16, device=dev) for _ in range(3)]
        _capi.lib()
        _capi._check(_capi.lib().jenga_sp_qkv_prologue(None, _capi._p(x), _capi._p(x), _capi._p(x), _capi._p(w[0]),
                                                       _capi._p(w[1]), _capi._p(w[2]), None, None, None, None, 1, 40, 8,
                                                       8, 2, 2, *_capi._bshd_strides(x), 0, *_capi._bshd_strides(w[0]),
                                                       0, 1e-6, 0), "jenga_sp_qkv_prologue")


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [(384, 512, 256), (1000, 768, 1024), (77, 3072, 512)])
def test_linear_epilogues_vs_fp32_reference(dev, dt, M, N, K):
    """jenga_linear (hipBLASLt GEMM + epilogue, fp32 accumulation, ONE rounding) against an fp32 torch reference of the
    same expression: plain, bias, tanh-GELU into a STRIDED destination (the MLP half of the single-stream blocks'
    concat buffer), and gate * (x W^T) + bias' + residual (apply_gate + residual add).  Tolerance: one rounding of the
    result to the 16-bit dtype + fp32 summation-order noise."""
    from jenga_amd import _capi
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    x = torch.randn(1, M, K + 64, generator=g, device=dev).to(dt)[..., :K]        # strided rows
    w = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(dt)
    b = (torch.randn(N, generator=g, device=dev) * 0.1).to(dt)
    xf, wf, bf = x.float(), w.float(), b.float()
    ulp = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11

    def close(got, ref):
        err = (got.float() - ref).abs()
        bound = ulp * ref.abs().clamp_min(2.0 ** -6) + 1e-3
        assert bool((err <= bound).all()), (err.max().item(), (err / bound).max().item())

    close(_capi.linear(x, w), xf @ wf.t())
    close(_capi.linear(x, w, b), xf @ wf.t() + bf)
    # GELU epilogue into the right part of a wider buffer; the left part must stay untouched
    cat = torch.full((1, M, 128 + N), 3.0, dtype=dt, device=dev)
    _capi.linear(x, w, b, act=_capi.ACT_GELU_TANH, out=cat[..., 128:])
    close(cat[..., 128:], torch.nn.functional.gelu(xf @ wf.t() + bf, approximate="tanh"))
    assert bool((cat[..., :128] == 3.0).all())
    # gate + residual: res + gate * (x W^T + b)  ==  gate * (x W^T) + (gate * b) + res
    gate = (torch.randn(1, N, generator=g, device=dev) * 0.5).to(dt)
    res = torch.randn(1, M, N, generator=g, device=dev).to(dt)
    got = _capi.linear(x, w, b * gate.reshape(-1), gate=gate, res=res)
    close(got, res.float() + gate.float() * (xf @ wf.t()) + (b * gate.reshape(-1)).float())
    # the blocks' helper (bias folded by the helper) against the unfused kernels it replaces: within the roundings the
    # eager chain adds (GEMM -> dtype, * gate -> dtype, + res -> dtype)
    from jenga_amd.dit import linear_gate_residual
    lin = torch.nn.Linear(K, N, dtype=dt, device=dev)
    with torch.no_grad():
        lin.weight.copy_(w)
        lin.bias.copy_(b)
        fused = linear_gate_residual(lin, x, gate, res)
        y = lin(x)
        unfused = _capi.gate_residual(res, y, gate)
    exact = res.float() + gate.float() * (xf @ wf.t() + bf)
    mant = 7 if dt == torch.bfloat16 else 10
    ulp_of = lambda t_: torch.exp2(torch.floor(torch.log2(t_.abs().clamp_min(2.0 ** -4))) - mant)
    # the fused call: fp32 all the way (the bias was folded as dtype(b * gate): one more rounding of a small term), then
    # ONE rounding of the result
    e_f = (fused.float() - exact).abs()
    assert bool((e_f <= 0.5 * ulp_of(exact) + ulp_of(bf * gate.float()) + 1e-3).all()), e_f.max().item()
    # the eager chain it replaces rounds the GEMM, the gate product and the sum: half an ulp of each TERM
    e_u = (unfused.float() - exact).abs()
    yg = y.float() * gate.float()
    # (an ulp of each, not half: values next to a power of two change their ulp by the rounding itself)
    assert bool((e_u <= ulp_of(y.float()) * gate.float().abs() + ulp_of(yg) + ulp_of(exact) + 2e-3).all()), e_u.max().item()
    assert float(e_f.mean()) <= float(e_u.mean())        # one rounding is closer to exact than three
    with pytest.raises(_capi.JengaError):
        _capi.linear(x, w, b, act=_capi.ACT_GELU_TANH, res=res)


@pytest.mark.parametrize("M,N,K", [(300, 512, 256), (1280, 1536, 1536), (257, 768, 3072)])
def test_linear_into_a_float32_residual_stream(dev, M, N, K):
    """jenga_linear with JENGA_OUT_F32 (round 6): res and out float32, x and w bf16 -- the Wan blocks' `x = x + y * e`
    (wan/modules/model_mul.py:334-341) in the epilogue of the GEMM that produces y.  Against the same expression in float64 on
    the bf16 operands: the result is ONE fp32 rounding of it plus the fp32 summation order of K products; in place (out is res)
    and into a new tensor; without gate (the cross-attention update) and without bias.  And against the unfused pair it
    replaces (16-bit GEMM output, then jenga_wan_gate_residual): that one carries the extra 16-bit rounding of y."""
    from jenga_amd import _capi
    from jenga_amd.wan_dit import _linear_into_stream
    g = torch.Generator(device=dev).manual_seed(M * 7 + N)
    a = torch.randn(1, M, K, generator=g, device=dev).to(torch.bfloat16)
    lin = torch.nn.Linear(K, N, dtype=torch.bfloat16, device=dev)
    with torch.no_grad():
        lin.weight.copy_((torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.bfloat16))
        lin.bias.copy_((torch.randn(N, generator=g, device=dev) * 0.1).to(torch.bfloat16))
    gate = torch.randn(1, N, generator=g, device=dev) * 0.7
    x = torch.randn(1, M, N, generator=g, device=dev)
    y64 = a.double() @ lin.weight.double().t() + lin.bias.double()
    for gt in (gate, None):
        exact = x.double() + (y64 * gt.double() if gt is not None else y64)
        keep = x.clone()
        got = _linear_into_stream(lin, a, gt, x)
        assert got.dtype == torch.float32 and got.data_ptr() != x.data_ptr() and torch.equal(x, keep)
        bound = 2.0 ** -22 * exact.abs().clamp_min(1.0) + 2e-5 * (K / 256) ** 0.5      # an fp32 rounding + summation noise
        err = (got.double() - exact).abs()
        assert bool((err <= bound).all()), (err.max().item(), (err / bound).max().item())
        x2 = x.clone()
        got2 = _linear_into_stream(lin, a, gt, x2, out=x2)
        assert got2.data_ptr() == x2.data_ptr() and torch.equal(x2, got)
        # the unfused pair: y rounded to bf16 first -- further from exact than the fused call, and within that rounding of it
        unfused = _capi.wan_gate_residual(x, lin(a), gt)
        e_u = (unfused.double() - exact).abs()
        assert float(err.mean()) <= float(e_u.mean())
        ulp_y = torch.exp2(torch.floor(torch.log2(y64.abs().clamp_min(2.0 ** -6))) - 7)
        assert bool(((got.double() - unfused.double()).abs() <= ulp_y * (gt.double().abs() if gt is not None else 1.0) + 1e-4).all())
    with pytest.raises((ValueError, _capi.JengaError)):
        _capi.linear(a, lin.weight, None, act=_capi.ACT_GELU_TANH, res=x)


def test_linear_choices_export_import_roundtrip(dev):
    """jenga_linear_export_choices / _import_choices (rank 0's hipBLASLt choices adopted by every rank: the replicated
    text stream must see the same arithmetic everywhere): importing a device's own choices reproduces its results bit
    for bit, and a forced choice is honoured without timing."""
    from jenga_amd import _capi
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn(1, 640, 512, generator=g, device=dev, dtype=torch.bfloat16)
    w = torch.randn(768, 512, generator=g, device=dev, dtype=torch.bfloat16) * 0.05
    b = torch.randn(768, generator=g, device=dev, dtype=torch.bfloat16)
    y0 = _capi.linear(x, w, b, act=_capi.ACT_GELU_TANH)
    rec = _capi.linear_export_choices()
    assert rec.dim() == 2 and rec.shape[1] == 12 and rec.shape[0] >= 1
    mine = [r for r in rec.tolist() if r[0] == 640 and r[1] == 768 and r[2] == 512]
    # (ABI 4) the last field is packed: list index | requested candidate count << 8 | (hipBLASLt solution index + 1) << 16
    packed = mine[0][11]
    choice, want, sol = packed & 0xff, (packed >> 8) & 0xff, (packed >> 16) - 1
    assert 0 <= choice < 32 and 1 <= want <= 32 and choice < want and sol >= 0, (choice, want, sol)
    before = _capi.linear_import_mismatches()
    _capi.linear_import_choices(rec)                     # drops the plans; the next call rebuilds them from the record
    y1 = _capi.linear(x, w, b, act=_capi.ACT_GELU_TANH)
    assert torch.equal(y0, y1)
    assert _capi.linear_import_mismatches() == before    # the rebuilt plan runs the exported solution
    rec2 = _capi.linear_export_choices()
    mine2 = [r for r in rec2.tolist() if r[0] == 640 and r[1] == 768 and r[2] == 512]
    assert mine2 and (mine2[0][11] >> 16) - 1 == sol
    with pytest.raises(_capi.JengaError):
        bad = rec.clone()
        bad[0, 11] = 99 | (8 << 8)
        _capi.linear_import_choices(bad)
    with pytest.raises(ValueError):
        _capi.linear_import_choices(torch.zeros(3, 5, dtype=torch.int64))


def test_wan_norm_rope_rejects_a_batch_with_rope_tables(dev):
    from jenga_amd import _capi
    x = torch.randn(2, 64, 256, device=dev).to(torch.bfloat16)
    wgt = torch.ones(256, device=dev)
    cs = torch.ones(64, 64, dtype=torch.float64, device=dev)
    with pytest.raises(ValueError):
        _capi.wan_norm_rope(x, wgt, cs, cs, 64, 1e-6)
    _capi.wan_norm_rope(x, wgt, None, None, 0, 1e-6)      # norm + cast only: any leading shape
    with pytest.raises(_capi.JengaError):                 # the C entry: s_rope beyond the rows of the one sequence
        _capi._check(_capi.lib().jenga_wan_norm_rope(None, _capi._p(x), _capi._p(x), _capi._p(wgt), _capi._p(cs),
                                                     _capi._p(cs), 32, 256, 256, 256, 64, 1e-6, 0), "jenga_wan_norm_rope")
