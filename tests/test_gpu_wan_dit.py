"""GPU (-m gpu): the Wan2.1 DiT harness -- the two fp32-residual glue kernels against the oracle,
WanAttentionBlock against tensors captured inside the reference's own block (tests/golden/wan_block_case.npz), and the
WanDiT forward (patchify, time/text embeddings, TeaCache driver, head, unpatchify) end to end on a small grid."""
import os
import sys

import numpy as np
import pytest
import torch

from helpers import assert_ulp_close, from_bits, to_np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import inputs  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("C,rows", [(256, 300), (1536, 257), (5120, 130)])
def test_wan_glue_kernels_vs_oracle(dev, C, rows):
    from jenga_amd import _capi
    from oracle import wan as ow
    g = torch.Generator().manual_seed(C)
    x = torch.randn(1, rows, C, generator=g) * 3 + 0.5
    w, b = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    sh, sc = 0.3 * torch.randn(1, C, generator=g), 0.3 * torch.randn(1, C, generator=g)
    xd = x.to(dev)
    for kw, okw in ((dict(shift=sh, scale=sc), (None, None, sh[0].numpy(), sc[0].numpy())),
                    (dict(weight=w, bias=b), (w.numpy(), b.numpy(), None, None)),
                    (dict(weight=w, bias=b, shift=sh, scale=sc), (w.numpy(), b.numpy(), sh[0].numpy(), sc[0].numpy()))):
        for rl in (False, True):
            got = _capi.wan_ln_modulate(xd, eps=1e-6, round_ln=rl, **{k: v.to(dev) for k, v in kw.items()})
            assert got.dtype == torch.bfloat16
            ref = ow.ln_modulate(x.numpy(), *okw, 1e-6, "bfloat16", round_ln=rl)
            # equal up to rare last-place flips; where shift cancels the scaled norm the result is tiny and an fp32-level
            # difference of the LayerNorm statistics is worth several of ITS ulps, so the ulp is floored at |t| = 2^-5
            # (and with round_ln a flip of the intermediate 16-bit rounding is scaled by 1 + scale: allow 3 such ulps)
            a, b_ = to_np(got), ref
            diff = a != b_
            assert diff.mean() <= 3e-3, diff.mean()
            if diff.any():
                mag = np.maximum(np.maximum(np.abs(a[diff]), np.abs(b_[diff])), 2.0 ** -5 if not rl else 1.0)
                ulp = np.exp2(np.floor(np.log2(mag)) - 7)
                assert (np.abs(a[diff] - b_[diff]) / ulp).max() <= (3.001 if rl else 1.001)
    y = (torch.randn(1, rows, C, generator=g)).to(torch.bfloat16)
    gate = torch.randn(1, C, generator=g)
    out = _capi.wan_gate_residual(xd, y.to(dev), gate.to(dev))
    assert out.dtype == torch.float32 and out.data_ptr() != xd.data_ptr()
    assert np.array_equal(out.cpu().numpy(), ow.gate_residual(x.numpy(), to_np(y), gate[0].numpy()))       # bit-exact
    x2 = xd.clone()
    _capi.wan_gate_residual(x2, y.to(dev), None, out=x2)                                                      # in place
    assert np.array_equal(x2.cpu().numpy(), ow.gate_residual(x.numpy(), to_np(y)))


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("C,rows,s_rope", [(1536, 300, 300), (5120, 130, 100), (256, 77, 77)])
def test_wan_norm_rope_equals_rmsnorm_rows_plus_rope_complex(dev, dt, C, rows, s_rope):
    """jenga_wan_norm_rope (one pass: WanRMSNorm with its fp32 weight + float64 RoPE + bf16 cast, into a block-padded
    buffer) against the two kernels it replaces -- jenga_rmsnorm_rows (fp32 out) + jenga_rope_complex (fp32 in, bf16
    out), themselves pinned to the reference goldens -- bit for bit; without tables: norm + cast."""
    from jenga_amd import _capi
    g = torch.Generator(device=dev).manual_seed(C + rows)
    x = (torch.randn(1, rows, C + 64, generator=g, device=dev) * 2).to(dt)[..., :C]           # strided rows
    w = 1 + 0.1 * torch.randn(C, generator=g, device=dev)
    H = C // 128
    ang = torch.randn(s_rope, 64, generator=g, device=dev, dtype=torch.float64)
    cos, sin = ang.cos(), ang.sin()
    ref = _capi.rope_complex(_capi.rmsnorm_rows(x, w, 1e-6).view(1, rows, H, 128), cos, sin, s_rope,
                             out_dtype=torch.bfloat16).view(1, rows, C)
    pad = torch.full((rows + 51, C), 5.0, dtype=torch.bfloat16, device=dev)                    # rows land in its prefix
    _capi.wan_norm_rope(x, w, cos, sin, s_rope, 1e-6, out=pad)
    torch.cuda.synchronize()
    assert torch.equal(pad[:rows], ref[0]) and bool((pad[rows:] == 5.0).all())
    plain = _capi.wan_norm_rope(x, w, None, None, 0, 1e-6)
    assert torch.equal(plain, _capi.rmsnorm_rows(x, w, 1e-6).to(torch.bfloat16))


def _block(dev):
    from jenga_amd.wan_dit import WanAttentionBlock
    from jenga_amd.modules.wan import wan_freqs
    c = inputs.WAN_BLOCK
    inp = inputs.wan_block_inputs()
    blk = WanAttentionBlock("t2v_cross_attn", c["dim"], c["ffn_dim"], c["num_heads"], qk_norm=True,
                            cross_attn_norm=True, eps=c["eps"], dtype=torch.bfloat16, device=dev)
    missing, unexpected = blk.load_state_dict(inp["state"], strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    return blk, inp, wan_freqs(c["dim"] // c["num_heads"]), c


@pytest.mark.parametrize("fuse", [False, True])
def test_wan_block_vs_reference_block(dev, fuse, monkeypatch):
    """Same weights, inputs and RoPE remap as the reference block run on CPU (dense path, sa_drop_rate = 0).  The
    captured intermediates are bf16: the kernels must agree to the ulp; the fp32 block output carries the hipBLASLt vs
    CPU GEMM summation-order noise of three bf16 branches (|y| ~ 1, one bf16 ulp = 2^-8).
    fuse=False: the bit-comparable configuration (gate + residual as their own kernel, every branch output captured);
    fuse=True (the default since round 6): gate + residual in the epilogue of the GEMM that produces the branch
    (jenga_linear, JENGA_OUT_F32) -- one rounding where the reference rounds the branch to 16 bits first, same bounds on the
    block output."""
    from jenga_amd import wan_dit
    monkeypatch.setattr(wan_dit, "WAN_FUSE_GATE", fuse)
    blk, inp, freqs, c = _block(dev)
    g = np.load(os.path.join(GOLD, "wan_block_case.npz"))
    f, h, w = c["grid"]
    cap = {}
    blk.self_attn.register_forward_hook(lambda m, a, o: cap.update(h1=a[0].clone(), y1=o.clone()))
    blk.cross_attn.register_forward_hook(lambda m, a, o: cap.update(h3=a[0].clone(), y3=o.clone()))
    ffn_hidden = blk.ffn_hidden          # (ffn[0] runs as a GEMM + GELU epilogue: capture its input at the method)
    blk.ffn_hidden = lambda h_: (cap.update(h2=h_.clone()), ffn_hidden(h_))[1]
    kw = dict(seq_lens=torch.tensor([f * h * w]), grid_sizes=torch.tensor([[f, h, w]]), freqs=freqs,
              context=inp["context"].to(dev), context_lens=None, sa_drop_rate=0.0, freq_remap=inp["remap"].to(dev),
              block_neighbor_list=None, p_remain_rates=0.8)
    for tag, x_in, was16 in (("first", inp["x"], True), ("later", inp["x"] * 1.7 + 0.123, False)):
        x_dev = x_in.to(dev)
        keep = x_dev.clone()
        y = blk(x_dev, inp["e"].to(dev), x_was_16bit=was16, **kw)
        assert y.dtype == torch.float32 and torch.equal(x_dev, keep), "the block must not modify its input"
        assert_ulp_close(to_np(cap["h1"]), to_np(from_bits(g[f"{tag}_h1"], "bfloat16")), "bfloat16", max_frac=3e-3)
        ref = g[f"{tag}_out"]
        err = np.abs(y.cpu().numpy() - ref)
        assert err.max() <= 6e-2 and err.mean() <= 4e-3, (tag, err.max(), err.mean())
        if tag == "later" and not fuse:          # (fused: the hooks see the branches BEFORE their output projection)
            for k_, tol in (("y1", 3e-2), ("y3", 3e-2)):
                d = np.abs(to_np(cap[k_]) - to_np(from_bits(g[f"later_{k_}"], "bfloat16")))
                assert d.max() <= tol, (k_, d.max())
            # the modulated norms downstream see slightly different residuals: compare loosely
            for k_ in ("h3", "h2"):
                d = np.abs(to_np(cap[k_]) - to_np(from_bits(g[f"later_{k_}"], "bfloat16")))
                assert d.max() <= 8e-2 and d.mean() <= 4e-3, (k_, d.max(), d.mean())


def test_wan_block_sparse_path_runs_and_matches_dense_when_everything_is_kept(dev):
    """sa_drop_rate > 0.25 takes the block-sparse kernels; with a drop rate that keeps every block (top_k = all) the
    result must equal the dense path's (same kernels underneath, dense == all blocks kept)."""
    blk, inp, freqs, c = _block(dev)
    f, h, w = c["grid"]
    kw = dict(seq_lens=torch.tensor([f * h * w]), grid_sizes=torch.tensor([[f, h, w]]), freqs=freqs,
              context=inp["context"].to(dev), context_lens=None, freq_remap=inp["remap"].to(dev), p_remain_rates=1.0)
    x = (inp["x"] * 1.7 + 0.123).to(dev)
    nbm = torch.ones(1, 1, dtype=torch.bool)
    dense = blk(x, inp["e"].to(dev), sa_drop_rate=0.0, block_neighbor_list=None, **kw)
    sparse = blk(x, inp["e"].to(dev), sa_drop_rate=0.3, block_neighbor_list=nbm, **kw)
    assert torch.equal(dense, sparse)


def test_wan_dit_forward_end_to_end(dev):
    """Small WanDiT: two CFG streams per step through the TeaCache driver, sliced-Gilbert reorder, head + unpatchify.
    Checks shapes, finiteness, that the reorder is transparent (identity curve == sliced curve up to the attention's
    block structure being dense here) and that a cached step replays the stored residual."""
    from jenga_amd import gilbert as G
    from jenga_amd.wan_dit import WanDiT
    torch.manual_seed(0)
    m = WanDiT(dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64, text_len=32, dtype=torch.bfloat16,
               device=dev)
    for p_ in m.parameters():
        if p_.dim() >= 2:
            torch.nn.init.normal_(p_, std=0.05)
    F_, H_, W_ = 3, 16, 16
    grid = (F_, H_ // 2, W_ // 2)
    L = grid[0] * grid[1] * grid[2]
    l2h, h2l = G.sliced_gilbert_mapping(*grid, as_tensor=True)
    nbm = G.sliced_gilbert_block_neighbor_mapping(*grid, as_tensor=True)
    x = [torch.randn(16, F_, H_, W_, device=dev)]
    ctx = [torch.randn(20, 64, device=dev)]
    t = torch.tensor([900.0], device=dev)
    outs = {}
    for name, (a, b) in (("curve", (l2h, h2l)), ("ident", (torch.arange(L, device=dev), torch.arange(L, device=dev)))):
        m.set_curve(a, b, nbm)
        m.enable_teacache(num_steps=4, thresh=0.0, task="t2v-1.3B", enable=False)
        y = m(x, t, ctx, seq_len=L, sa_drop_rate=0.0)[0]
        assert y.shape == (16, F_, H_, W_) and y.dtype == torch.float32 and torch.isfinite(y).all()
        outs[name] = y
    # dense attention is permutation-equivariant: the curve only changes summation order inside the kernels
    assert (outs["curve"] - outs["ident"]).abs().max().item() <= 5e-2 * outs["ident"].abs().max().item()
    # TeaCache: with a huge threshold the third call of a stream (cnt = 4, even) is skipped and replays the residual
    m.set_curve(l2h, h2l, nbm)
    m.enable_teacache(num_steps=6, thresh=1e9, task="t2v-1.3B", enable=True)
    calls = []
    for i in range(6):
        before = m.tea.cnt
        m(x, torch.tensor([900.0 - 10 * (i // 2)], device=dev), ctx, seq_len=L, sa_drop_rate=0.0)
        calls.append((before, m.tea.residual[before % 2] is not None))
    assert m.tea.cnt == 6 and all(r for _, r in calls)


def test_wan_dit_forward_vs_reference_model_and_teacache(dev):
    """WanDiT against the reference WanModel driven by the reference's own `teacache_forward` (jenga_wan.py) on CPU:
    same parameters (reference state-dict names; the Conv3d patch kernel flattened), 7 steps x 2 CFG streams with
    timesteps chosen so that the TeaCache rule skips some forwards.  The skip decisions must be identical, the outputs
    equal up to bf16 GEMM noise through two blocks."""
    from jenga_amd import gilbert as G
    from jenga_amd.wan_dit import WanDiT
    c = inputs.WAN_MODEL
    inp = inputs.wan_model_inputs()
    g = np.load(os.path.join(GOLD, "wan_forward_case.npz"))
    m = WanDiT(text_len=c["text_len"], in_dim=c["in_dim"], dim=c["dim"], ffn_dim=c["ffn_dim"], freq_dim=c["freq_dim"],
               text_dim=c["text_dim"], out_dim=c["out_dim"], num_heads=c["num_heads"], num_layers=c["num_layers"],
               cross_attn_norm=True, dtype=torch.bfloat16, device=dev)
    sd = {}
    for k_, v_ in m.state_dict().items():
        if k_ == "patch_embedding.weight":
            sd[k_] = inputs.wan_param(k_, (c["dim"], c["in_dim"], 1, 2, 2)).reshape(c["dim"], -1)
        else:
            sd[k_] = inputs.wan_param(k_, tuple(v_.shape))
    m.load_state_dict(sd, strict=True)
    F_, H_, W_ = c["latent"]
    grid = (F_, H_ // 2, W_ // 2)
    L = grid[0] * grid[1] * grid[2]
    l2h, h2l = G.sliced_gilbert_mapping(*grid, as_tensor=True)
    m.set_curve(l2h, h2l, G.sliced_gilbert_block_neighbor_mapping(*grid, as_tensor=True))
    m.enable_teacache(num_steps=c["steps"], thresh=c["thresh"], task="t2v-1.3B", use_ret_steps=False, enable=True)
    used_cache = []
    call = 0
    for i, t in enumerate(inp["timesteps"]):
        for j, ctx in enumerate(inp["context"]):
            y = m([inp["x"].to(dev)], torch.tensor([t], device=dev), [ctx.to(dev)], seq_len=L, sa_drop_rate=0.0)[0]
            used_cache.append(not m.last_computed)
            ref = g[f"out_{i}_{j}"]
            err = np.abs(y.cpu().numpy() - ref)
            assert err.max() <= 4e-2 * max(1.0, np.abs(ref).max()) and err.mean() <= 4e-3, (i, j, err.max(), err.mean())
            call += 1
    assert used_cache == g["used_cache"].tolist(), (used_cache, g["used_cache"].tolist())
    assert any(used_cache) and not all(used_cache)


def test_wan_1p3b_full_model_vs_reference_cpu_forward(dev):
    """BASELINE.json configs[0]: the full Wan2.1-1.3B architecture (30 layers, dim 1536, 12 heads, ffn 8960, 512 x 4096
    text) on the 256x256x17f latent (1280 tokens, dense attention) against ONE forward of the reference model through
    its own Jenga `teacache_forward` on the CPU (tests/golden/make_golden.py gen_wan_1p3b, 45 s there).  The 1.4 G
    parameters are regenerated from their names.  Output |y| <= 3.5; the bound covers bf16 GEMM summation-order noise
    through 30 blocks."""
    from jenga_amd import gilbert as G
    from jenga_amd.wan_dit import WanDiT
    c = inputs.WAN_1P3B
    inp = inputs.wan_1p3b_inputs()
    g = np.load(os.path.join(GOLD, "wan_1p3b_forward.npz"))
    m = WanDiT(text_len=c["text_len"], in_dim=c["in_dim"], dim=c["dim"], ffn_dim=c["ffn_dim"], freq_dim=c["freq_dim"],
               text_dim=c["text_dim"], out_dim=c["out_dim"], num_heads=c["num_heads"], num_layers=c["num_layers"],
               cross_attn_norm=True, dtype=torch.bfloat16, device=dev)
    with torch.no_grad():
        for k_, p_ in m.state_dict().items():
            if k_ == "patch_embedding.weight":
                w = inputs.wan_param(k_, (c["dim"], c["in_dim"], 1, 2, 2), fan_in_gain=c["gain"]).reshape(c["dim"], -1)
            else:
                w = inputs.wan_param(k_, tuple(p_.shape), fan_in_gain=c["gain"])
            p_.copy_(w.to(device=dev, dtype=p_.dtype))
    F_, H_, W_ = c["latent"]
    grid = (F_, H_ // 2, W_ // 2)
    L = grid[0] * grid[1] * grid[2]
    l2h, h2l = G.sliced_gilbert_mapping(*grid, as_tensor=True)
    m.set_curve(l2h, h2l, G.sliced_gilbert_block_neighbor_mapping(*grid, as_tensor=True))
    m.enable_teacache(num_steps=10, thresh=0.15, task="t2v-1.3B", use_ret_steps=False, enable=True)
    y = m([inp["x"].to(dev)], torch.tensor([c["timestep"]], device=dev), [inp["context"].to(dev)], seq_len=L,
          sa_drop_rate=0.0)[0]
    ref = g["out"]
    got = y.cpu().numpy()
    assert got.shape == ref.shape == (16, 5, 32, 32) and np.isfinite(got).all()
    err = np.abs(got - ref)
    print("wan-1.3B full forward: max abs err %.4f, mean abs err %.5f at |ref| max %.2f mean %.3f" % (err.max(), err.mean(), np.abs(ref).max(), np.abs(ref).mean()))
    assert err.max() <= 3e-2 and err.mean() <= 4e-3, (err.max(), err.mean(), np.abs(ref).max(), np.abs(ref).mean())


@pytest.mark.parametrize("dt,Sq,Skv,H,kv_len", [(torch.bfloat16, 384, 512, 3, 512), (torch.float16, 128, 128, 2, 128),
                                                 (torch.bfloat16, 1280, 512, 12, 512), (torch.bfloat16, 256, 128, 2, 24),
                                                 (torch.float16, 128, 256, 3, 200), (torch.bfloat16, 128, 384, 2, 320)])
def test_cross_attention_dense_mode_vs_oracle(dev, dt, Sq, Skv, H, kv_len):
    """jenga_cross_attn_fwd (WanT2VCrossAttention, model_mul.py:183-205: flash_attention over the 512 context tokens,
    k_lens = None) = the LP kernel's text-row mode with a kv sequence of its own length: against the oracle's dense
    rows (oracle.attention.text_rows: fp32 scores x d^-0.5, natural softmax, P rounded to dtype before P.V) and against
    the same rows computed as text rows of jenga_bsattn_fwd (bit for bit: same kernel, same tiles)."""
    from jenga_amd import _capi
    from oracle import attention as oa
    g = torch.Generator().manual_seed(Sq + Skv)
    q = (torch.randn(1, Sq, H, 128, generator=g) * 1.5).to(dt)
    k = (torch.randn(1, Skv, H, 128, generator=g) * 1.5).to(dt)
    v = torch.randn(1, Skv, H, 128, generator=g).to(dt)
    # strided inputs: q as a slice of a wider buffer
    wide = torch.zeros(1, Sq, H + 1, 128, dtype=dt)
    wide[:, :, :H] = q
    # keys >= kv_len (inside the last block) are masked whatever the padded buffers hold there: poison them
    kd, vd = k.clone(), v.clone()
    kd[:, kv_len:] = 50.0
    vd[:, kv_len:] = -1000.0
    o = _capi.cross_attn_fwd(wide.to(dev)[:, :, :H], kd.to(dev), vd.to(dev), kv_len=kv_len)
    torch.cuda.synchronize()
    name = "bfloat16" if dt == torch.bfloat16 else "float16"
    tr = lambda t: to_np(t).transpose(0, 2, 1, 3)
    ref = oa.text_rows(tr(q), tr(k[:, :kv_len]), tr(v[:, :kv_len]), 128 ** -0.5, name).transpose(0, 2, 1, 3)
    err = np.abs(to_np(o) - ref)
    tol = 2e-2 if dt == torch.bfloat16 else 4e-3
    assert err.max() <= tol, err.max()
    assert np.median(err) <= tol / 10
    with pytest.raises(ValueError):
        _capi.cross_attn_fwd(q.to(dev)[:, :100], k.to(dev), v.to(dev))
    # the same rows as text rows of the block-sparse entry: [kv | q] sequence, nq_img = Skv / 128 image blocks whose lists
    # are irrelevant here, the q rows as "text" rows that see every key -- only comparable when the key sets agree, i.e.
    # K_all = [k | q-as-keys]; so compare on the reduced problem where the key sequence IS k followed by nothing:
    if Sq == Skv == kv_len:
        S = Sq
        vt = _capi.pack_v(v.to(dev), S // 128)
        o2 = _capi.bsattn_fwd(q.to(dev), k.to(dev), vt, None, None, None, 0, 128 ** -0.5, 0.0, S // 128)
        assert torch.equal(o2, o)


def test_cross_attention_at_the_wan14b_shape_sampled_rows(dev):
    """BASELINE.json configs[3]'s cross-attention shape: 75 600 query tokens (padded to 591 blocks) x 40 heads against the
    512-token context.  Size-independent property: V == 1 -> every output element of every row is 1 (the softmax weights
    of each of the 3 million rows sum to one); then 64 sampled (row, head) pairs against the oracle's dense rows."""
    from jenga_amd import _capi
    from oracle import attention as oa
    L, Lp, Lc, H = 75600, 75648, 512, 40
    g = torch.Generator(device=dev).manual_seed(14)
    q = torch.zeros(1, Lp, H, 128, device=dev, dtype=torch.bfloat16)
    q[:, :L] = (torch.randn(1, L, H, 128, generator=g, device=dev) * 1.3).to(torch.bfloat16)
    k = (torch.randn(1, Lc, H, 128, generator=g, device=dev) * 1.3).to(torch.bfloat16)
    ones = torch.ones(1, Lc, H, 128, device=dev, dtype=torch.bfloat16)
    o1 = _capi.cross_attn_fwd(q, k, ones)
    assert torch.all((o1[:, :L].float() - 1).abs() <= 2 ** -7), (o1[:, :L].float() - 1).abs().max().item()
    del o1, ones
    v = torch.randn(1, Lc, H, 128, generator=g, device=dev).to(torch.bfloat16)
    o = _capi.cross_attn_fwd(q, k, v)
    torch.cuda.synchronize()
    rows = torch.randint(0, L, (64,), generator=torch.Generator().manual_seed(1)).tolist() + [0, L - 1]
    heads = [(7 * i) % H for i in range(len(rows))]
    kk = k.float().cpu().numpy().transpose(0, 2, 1, 3)
    vv = v.float().cpu().numpy().transpose(0, 2, 1, 3)
    for r, h in zip(rows, heads):
        ref = oa.text_rows(q[:, r:r + 1, h:h + 1].float().cpu().numpy().transpose(0, 2, 1, 3), kk[:, h:h + 1],
                           vv[:, h:h + 1], 128 ** -0.5, "bfloat16")
        got = o[0, r, h].float().cpu().numpy()
        assert np.abs(got - ref[0, 0, 0]).max() <= 2e-2, (r, h, np.abs(got - ref[0, 0, 0]).max())
