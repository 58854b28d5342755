"""GPU (-m gpu): the DiT harness (caller rows a12/a14) -- block plumbing against an oracle composition, the Jenga
forward's gather/scatter + skip cache, and the sequence-parallel path over RCCL with world size 1."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import to_np

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _tiny_model(dev, depth=(1, 1), seed=0):
    from jenga_amd import dit
    dit.HUNYUAN_VIDEO_CONFIG["tiny"] = dict(mm_double_blocks_depth=depth[0], mm_single_blocks_depth=depth[1],
                                            rope_dim_list=[16, 56, 56], hidden_size=256, heads_num=2,
                                            mlp_width_ratio=4, guidance_embed=True)
    m = dit.JengaHYVideoDiT(config="tiny", text_states_dim=64, text_states_dim_2=32, dtype=torch.bfloat16, device=dev)
    return m.init_synthetic_weights(0.05, seed=seed)


def _inputs(dev, latent=(4, 16, 32)):
    g = torch.Generator(device=dev).manual_seed(7)
    x = torch.randn(1, 16, *latent, generator=g, device=dev, dtype=torch.bfloat16)
    text = torch.randn(1, 256, 64, generator=g, device=dev, dtype=torch.bfloat16)
    text2 = torch.randn(1, 32, generator=g, device=dev, dtype=torch.bfloat16)
    mask = torch.zeros(1, 256, dtype=torch.int64, device=dev)
    mask[:, :70] = 1
    return x, text, text2, mask


def _oracle_attention(q, k, v, top_k, seqlen, tb, amp, p, nbm):
    from oracle import attention as oa
    S = q.shape[1]
    cu = np.array([0, seqlen, S], np.int64)
    o = oa.block_sparse_attention(to_np(q), to_np(k), to_np(v), top_k, "bfloat16", cu_seqlens_q=cu, text_blocks=tb,
                                  text_amp=amp, block_neighbor_list=nbm, p_remain_rates=p)
    return torch.from_numpy(o).to(torch.bfloat16)


def test_single_stream_block_vs_oracle_composition(dev):
    """Same torch GEMMs on the GPU, hot-path ops replaced by the oracle: checks strided q/k/v views, RoPE on image
    tokens only, the write into the left part of the concat buffer, top_k truncation."""
    from jenga_amd import dit
    from oracle import gilbert as og
    from oracle import norm_rope as onr
    m = _tiny_model(dev)
    blk = m.single_blocks[0]
    S_img, S_txt, C, H = 512, 256, 256, 2
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn(1, S_img + S_txt, C, generator=g, device=dev, dtype=torch.bfloat16)
    vec = torch.randn(1, C, generator=g, device=dev, dtype=torch.bfloat16)
    nbm = og.gilbert_block_neighbor_mapping(2, 8, 32, 128)
    cos, sin = onr.rope_tables([16, 56, 56], [2, 8, 32], 256.0)
    cos_t, sin_t = torch.from_numpy(cos).to(dev), torch.from_numpy(sin).to(dev)
    cu = torch.tensor([0, S_img + 70, S_img + S_txt], dtype=torch.int32, device=dev)
    curve = [[None, None, torch.from_numpy(nbm).to(dev)]]
    out = blk(x, vec, S_txt, cu, cu, S_img + S_txt, S_img + S_txt, (cos_t, sin_t), 0.5, 0.3, curve, 0.3)
    # ---- oracle composition
    torch.set_grad_enabled(False)
    shift, scale, gate = blk.modulation(vec).chunk(3, dim=-1)
    lin1 = blk.linear1(dit.modulate(blk.pre_norm(x), shift, scale))
    qkv = lin1[..., :3 * C].reshape(1, S_img + S_txt, 3, H, 128)
    q = onr.rmsnorm(to_np(qkv[:, :, 0]), to_np(blk.q_norm.weight), "bfloat16")
    k = onr.rmsnorm(to_np(qkv[:, :, 1]), to_np(blk.k_norm.weight), "bfloat16")
    q[:, :S_img] = onr.apply_rotary_emb(q[:, :S_img], cos, sin, "bfloat16")
    k[:, :S_img] = onr.apply_rotary_emb(k[:, :S_img], cos, sin, "bfloat16")
    attn = _oracle_attention(torch.from_numpy(q), torch.from_numpy(k), qkv[:, :, 2].cpu(), int(0.5 * 4), S_img + 70, 2,
                             0.3, 0.3, nbm).to(dev)
    cat = torch.cat((attn, F.gelu(lin1[..., 3 * C:], approximate="tanh")), 2)
    branch = blk.linear2(cat) * gate.unsqueeze(1)
    ref = x + branch
    err = (out.float() - ref.float()).abs()
    # residual stream values reach |x| ~ 8 where one bf16 ulp is 0.0625.  The eager chain rounds the GEMM, the gate
    # product and the sum; the block computes gate * GEMM + residual in the GEMM epilogue with ONE rounding (round 3), so
    # the two differ by the roundings of the TERMS: bound = 2 ulp of the largest of |x|, |branch|, |result| + 0.02
    mag = torch.maximum(torch.maximum(ref.float().abs(), x.float().abs()), branch.float().abs())
    bound = 2 * torch.exp2(torch.floor(torch.log2(mag.clamp_min(1e-3))) - 7) + 0.02
    assert (err <= bound).all() and err.mean().item() <= 2e-3, (err.max().item(), err.mean().item())


def test_double_stream_block_vs_oracle_composition(dev):
    from jenga_amd import dit
    from oracle import gilbert as og
    from oracle import norm_rope as onr
    m = _tiny_model(dev)
    blk = m.double_blocks[0]
    S_img, S_txt, C, H = 512, 256, 256, 2
    g = torch.Generator(device=dev).manual_seed(4)
    img = torch.randn(1, S_img, C, generator=g, device=dev, dtype=torch.bfloat16)
    txt = torch.randn(1, S_txt, C, generator=g, device=dev, dtype=torch.bfloat16)
    vec = torch.randn(1, C, generator=g, device=dev, dtype=torch.bfloat16)
    nbm = og.gilbert_block_neighbor_mapping(2, 8, 32, 128)
    cos, sin = onr.rope_tables([16, 56, 56], [2, 8, 32], 256.0)
    cos_t, sin_t = torch.from_numpy(cos).to(dev), torch.from_numpy(sin).to(dev)
    cu = torch.tensor([0, S_img + 70, S_img + S_txt], dtype=torch.int32, device=dev)
    curve = [[None, None, torch.from_numpy(nbm).to(dev)]]
    o_img, o_txt = blk(img, txt, vec, cu, cu, S_img + S_txt, S_img + S_txt, (cos_t, sin_t), 0.5, 0.3, curve, 0.3)
    torch.set_grad_enabled(False)
    im = blk.img_mod(vec).chunk(6, dim=-1)
    tm = blk.txt_mod(vec).chunk(6, dim=-1)
    iqkv = blk.img_attn_qkv(dit.modulate(blk.img_norm1(img), im[0], im[1])).view(1, S_img, 3, H, 128)
    tqkv = blk.txt_attn_qkv(dit.modulate(blk.txt_norm1(txt), tm[0], tm[1])).view(1, S_txt, 3, H, 128)
    nq = lambda t, w: onr.rmsnorm(to_np(t), to_np(w), "bfloat16")
    q = np.concatenate([onr.apply_rotary_emb(nq(iqkv[:, :, 0], blk.img_attn_q_norm.weight), cos, sin, "bfloat16"),
                        nq(tqkv[:, :, 0], blk.txt_attn_q_norm.weight)], axis=1)
    k = np.concatenate([onr.apply_rotary_emb(nq(iqkv[:, :, 1], blk.img_attn_k_norm.weight), cos, sin, "bfloat16"),
                        nq(tqkv[:, :, 1], blk.txt_attn_k_norm.weight)], axis=1)
    v = torch.cat((iqkv[:, :, 2], tqkv[:, :, 2]), dim=1).cpu()
    attn = _oracle_attention(torch.from_numpy(q), torch.from_numpy(k), v, 2, S_img + 70, 2, 0.3, 0.3, nbm).to(dev)
    b1_img = blk.img_attn_proj(attn[:, :S_img]) * im[2].unsqueeze(1)
    m_img = img + b1_img
    b2_img = blk.img_mlp(dit.modulate(blk.img_norm2(m_img), im[3], im[4])) * im[5].unsqueeze(1)
    r_img = m_img + b2_img
    b1_txt = blk.txt_attn_proj(attn[:, S_img:]) * tm[2].unsqueeze(1)
    m_txt = txt + b1_txt
    b2_txt = blk.txt_mlp(dit.modulate(blk.txt_norm2(m_txt), tm[3], tm[4])) * tm[5].unsqueeze(1)
    r_txt = m_txt + b2_txt
    for got, ref, terms in ((o_img, r_img, (img, b1_img, m_img, b2_img)), (o_txt, r_txt, (txt, b1_txt, m_txt, b2_txt))):
        err = (got.float() - ref.float()).abs()
        # gate * GEMM + residual is ONE rounding in the block (GEMM epilogue) and three in the eager chain: the bound is
        # two bf16 ulps of the largest term that was rounded on the way + 0.03
        mag = ref.float().abs()
        for t_ in terms:
            mag = torch.maximum(mag, t_.float().abs())
        bound = 2 * torch.exp2(torch.floor(torch.log2(mag.clamp_min(1e-3))) - 7) + 0.03
        assert (err <= bound).all() and err.mean().item() <= 3e-3, (err.max().item(), err.mean().item())


def test_forward_gather_scatter_and_skip_cache(dev):
    m = _tiny_model(dev)
    x, text, text2, mask = _inputs(dev)
    cos, sin = m.set_stage((4, 16, 32), dev)
    l2h, h2l = m.linear_to_hilbert, m.hilbert_order
    assert torch.equal(h2l[l2h], torch.arange(l2h.numel(), device=dev))
    m.sa_drop_rate, m.text_amp, m.p_remain_rates = 0.5, 0.0, 0.3
    t = torch.tensor([900.0], device=dev)
    gd = torch.tensor([6000.0], device=dev)
    m.cnt = 0
    y0 = m(x, t, text, mask, text2, cos, sin, gd, return_dict=False)       # step 0: computed
    assert y0.shape == x.shape and torch.isfinite(y0.float()).all()
    res = m.previous_residual.clone()
    m.cnt = 5
    y1 = m(x, t, text, mask, text2, cos, sin, gd, return_dict=False)       # step 5: skipped -> cached residual
    assert torch.equal(m.previous_residual, res)
    assert (y1.float() - y0.float()).abs().max().item() < 0.05             # same input, same residual (bf16 re-rounding)


def test_row_kernels_on_the_side_stream_are_bit_identical_to_the_one_stream_order(dev):
    """Round 6: the HBM-bound row kernels in front of the attention launch run on a second stream beside the block's
    independent GEMM (jenga_amd.dit.ROWOPS_OVERLAP).  Same kernels, same inputs, same order of dependent work: the DiT forward
    (2 double-stream + 2 single-stream blocks, computed steps at two drop rates) must give the same bits as with the switch
    off -- repeated, so that a missing stream dependency (a race) has several chances to show."""
    from jenga_amd import dit
    m = _tiny_model(dev, depth=(2, 2), seed=3)
    x, text, text2, mask = _inputs(dev, latent=(4, 16, 64))               # 2048 image tokens = 16 blocks
    cos, sin = m.set_stage((4, 16, 64), dev)
    t = torch.tensor([900.0], device=dev)
    gd = torch.tensor([6000.0], device=dev)
    old = dit.ROWOPS_OVERLAP
    outs = {}
    try:
        for mode in (False, True, True, True):
            dit.ROWOPS_OVERLAP = mode
            for rate in (0.5, 0.75):
                m.sa_drop_rate, m.text_amp, m.p_remain_rates = rate, 0.2, 0.3
                m.cnt = 0
                y = m(x, t, text, mask, text2, cos, sin, gd, return_dict=False)
                torch.cuda.synchronize()
                if (rate,) not in outs:
                    assert mode is False
                    outs[(rate,)] = y.clone()
                else:
                    assert torch.equal(y, outs[(rate,)]), f"overlap={mode} rate={rate}: differs from the one-stream order"
    finally:
        dit.ROWOPS_OVERLAP = old
    assert torch.isfinite(outs[(0.5,)].float()).all() and not torch.equal(outs[(0.5,)], outs[(0.75,)])


def test_sequence_parallel_path_world1_rccl(dev):
    """World size 1 over RCCL: exercises the HIP head pack/unpack kernels, all_to_all_single / all_gather plumbing
    and the SP branches of blocks and driver; must reproduce the single-GPU path bit for bit."""
    import torch.distributed as dist
    from jenga_amd.modules import ulysses
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        m = _tiny_model(dev)
        x, text, text2, mask = _inputs(dev)
        cos, sin = m.set_stage((4, 16, 32), dev)
        m.sa_drop_rate, m.text_amp, m.p_remain_rates, m.enable_skip = 0.5, 0.2, 0.3, False
        t = torch.tensor([500.0], device=dev)
        gd = torch.tensor([6000.0], device=dev)
        ref = m(x, t, text, mask, text2, cos, sin, gd, return_dict=False)
        ulysses.init_sequence_parallel()
        for b in list(m.double_blocks) + list(m.single_blocks):
            b.hybrid_seq_parallel_attn = ulysses.UlyssesAttenCarve()
        got = m(x, t, text, mask, text2, cos, sin, gd, return_dict=False)
        assert torch.equal(got, ref)
    finally:
        dist.destroy_process_group()


def test_wan_teacache_forward_gather_scatter_and_cache(dev):
    """Wan driver skeleton: pad -> sliced-Hilbert gather -> blocks / cached residual -> scatter."""
    from jenga_amd import gilbert as G
    from jenga_amd.wan_driver import TeaCache, teacache_forward
    grid = (3, 6, 8)
    L = grid[0] * grid[1] * grid[2]
    seq_len = 160                                             # padded like the reference's seq_len argument
    l2h, h2l = G.sliced_gilbert_mapping(*grid, as_tensor=True)
    pad = torch.arange(L, seq_len, device=dev)
    order = torch.cat([h2l, pad])                             # padding tokens stay at the tail
    inv = torch.cat([l2h, pad])
    x = torch.randn(1, L, 64, device=dev).to(torch.bfloat16)
    e = torch.randn(1, 32, device=dev)
    tea = TeaCache(num_steps=4, thresh=1e9)                   # everything after the retained steps is skipped
    add_one = lambda t, **kw: t + 1
    outs = []
    for _ in range(6):
        y, calc = teacache_forward(x, e, e.unsqueeze(1), [add_one, add_one], tea, order, inv, seq_len=seq_len)
        outs.append((y, calc))
    xp = torch.cat([x, x.new_zeros(1, seq_len - L, 64)], 1)
    computed = (xp + 1) + 1                                   # two blocks, each result rounded to bf16
    cached = xp + (computed - xp)                             # skipped calls add the cached residual
    assert [c for _, c in outs] == [True, True, False, False, False, False]
    for y, calc in outs:
        assert y.shape == (1, seq_len, 64) and torch.equal(y, computed if calc else cached)


def test_i2v_token_replace_block_and_forward(dev):
    """HunyuanVideo-I2V flavour (hyvideo_i2v/modules/models_mul.py:393-506, jenga_hyi2v.py:79-92, 124-130): first-frame
    tokens take shift/scale/gate from the timestep-0 vector, four text blocks.  Block against an eager per-row
    composition + the oracle's attention; model forward: mask in curve order, txt_block_num, finite output that differs
    from the T2V forward."""
    from jenga_amd import dit
    from oracle import gilbert as og
    from oracle import norm_rope as onr
    m = _tiny_model(dev)
    blk = m.single_blocks[0]
    S_img, S_txt, C, H = 512, 512, 256, 2
    g = torch.Generator(device=dev).manual_seed(11)
    x = torch.randn(1, S_img + S_txt, C, generator=g, device=dev, dtype=torch.bfloat16)
    vec = torch.randn(1, C, generator=g, device=dev, dtype=torch.bfloat16)
    trv = torch.randn(1, C, generator=g, device=dev, dtype=torch.bfloat16)
    ffm = torch.rand(S_img, generator=g, device=dev) < 0.25
    nbm = og.gilbert_block_neighbor_mapping(2, 8, 32, 128)
    cos, sin = onr.rope_tables([16, 56, 56], [2, 8, 32], 256.0)
    cos_t, sin_t = torch.from_numpy(cos).to(dev), torch.from_numpy(sin).to(dev)
    cu = torch.tensor([0, S_img + 300, S_img + S_txt], dtype=torch.int32, device=dev)
    curve = [[None, None, torch.from_numpy(nbm).to(dev)]]
    out = blk(x, vec, S_txt, cu, cu, S_img + S_txt, S_img + S_txt, (cos_t, sin_t), 0.5, 0.3, curve, 0.3,
              txt_block_num=4, token_replace_vec=trv, first_frame_mask=ffm)
    torch.set_grad_enabled(False)
    full = torch.cat([ffm, torch.zeros(S_txt, dtype=torch.bool, device=dev)])[None, :, None]
    a, b = blk.modulation(vec).chunk(3, dim=-1), blk.modulation(trv).chunk(3, dim=-1)
    pre = blk.pre_norm(x)
    lin1 = blk.linear1(torch.where(full, dit.modulate(pre, b[0], b[1]), dit.modulate(pre, a[0], a[1])))
    qkv = lin1[..., :3 * C].reshape(1, S_img + S_txt, 3, H, 128)
    q = onr.rmsnorm(to_np(qkv[:, :, 0]), to_np(blk.q_norm.weight), "bfloat16")
    k = onr.rmsnorm(to_np(qkv[:, :, 1]), to_np(blk.k_norm.weight), "bfloat16")
    q[:, :S_img] = onr.apply_rotary_emb(q[:, :S_img], cos, sin, "bfloat16")
    k[:, :S_img] = onr.apply_rotary_emb(k[:, :S_img], cos, sin, "bfloat16")
    attn = _oracle_attention(torch.from_numpy(q), torch.from_numpy(k), qkv[:, :, 2].cpu(), int(0.5 * 4), S_img + 300, 4,
                             0.3, 0.3, nbm).to(dev)
    y = blk.linear2(torch.cat((attn, F.gelu(lin1[..., 3 * C:], approximate="tanh")), 2))
    branch = torch.where(full, y * b[2].unsqueeze(1), y * a[2].unsqueeze(1))
    ref = x + branch
    err = (out.float() - ref.float()).abs()
    # two bf16 ulps of the largest term (the block's GELU rides in the GEMM epilogue: one rounding where eager has two)
    mag = torch.maximum(torch.maximum(ref.float().abs(), x.float().abs()), branch.float().abs())
    bound = 2 * torch.exp2(torch.floor(torch.log2(mag.clamp_min(1e-3))) - 7) + 0.02
    assert (err <= bound).all() and err.mean().item() <= 2e-3, (err.max().item(), err.mean().item())
    # ---- model forward
    x_lat, _, text2, _ = _inputs(dev)
    text = torch.randn(1, 512, 64, generator=g, device=dev, dtype=torch.bfloat16)
    mask = torch.zeros(1, 512, dtype=torch.int64, device=dev)
    mask[:, :300] = 1
    cos_m, sin_m = m.set_stage((4, 16, 32), dev)
    m.sa_drop_rate, m.text_amp, m.p_remain_rates, m.enable_skip = 0.5, 0.0, 0.3, False
    t, gd = torch.tensor([900.0], device=dev), torch.tensor([6000.0], device=dev)
    seen = {}
    orig = m.single_blocks[0].forward

    def spy(*a_, **kw):
        seen.update(kw)
        return orig(*a_, **kw)

    m.single_blocks[0].forward = spy
    y_t2v = m(x_lat, t, text, mask, text2, cos_m, sin_m, gd, return_dict=False)
    assert seen["token_replace_vec"] is None and seen["txt_block_num"] == 4
    m.i2v_condition_type = "token_replace"
    y_i2v = m(x_lat, t, text, mask, text2, cos_m, sin_m, gd, return_dict=False)
    th_tw = (16 // 2) * (32 // 2)
    expect = torch.zeros(4 * th_tw, dtype=torch.bool, device=dev)
    expect[:th_tw] = True
    assert torch.equal(seen["first_frame_mask"], expect[m.hilbert_order]) and seen["token_replace_vec"] is not None
    assert torch.isfinite(y_i2v.float()).all() and not torch.equal(y_i2v, y_t2v)
    m.i2v_condition_type = None


def test_hy_blocks_vs_reference_blocks(dev):
    """MMSingleStreamBlock / MMDoubleStreamBlock against the reference's own blocks run on CPU in fp16 with the Jenga
    path on (tests/golden/make_golden.py gen_hy_blocks: real neighbours, block selection, the Triton kernel under the
    interpreter).  Same state dict (the parameter names are the reference's), same inputs.  fp16 outputs at |x| <= ~6:
    the bound is two fp16 ulps of the value plus the GEMM summation-order noise of the block's branches."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import inputs
    from jenga_amd import dit
    from jenga_amd.modules.posemb_layers import get_nd_rotary_pos_embed
    c = inputs.HY_BLOCK
    inp = inputs.hy_block_inputs()
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hy_blocks_case.npz"))
    dt = torch.float16
    nbm = torch.from_numpy(g["neighbors"]).to(dev)
    curve = [[None, None, nbm]]
    cos, sin = get_nd_rotary_pos_embed([16, 56, 56], list(c["grid"]), theta=256, use_real=True, theta_rescale_factor=1)
    cos, sin = cos.to(dev), sin.to(dev)
    S = inp["S_img"] + c["s_txt"]
    cu = inp["cu"].to(dev)
    sb = dit.MMSingleStreamBlock(c["hidden"], c["heads"], c["mlp_ratio"], dtype=dt, device=dev)
    missing, unexpected = sb.load_state_dict(inp["single"], strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    db = dit.MMDoubleStreamBlock(c["hidden"], c["heads"], c["mlp_ratio"], dtype=dt, device=dev)
    missing, unexpected = db.load_state_dict(inp["double"], strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    y = sb(inp["x"].to(dev), inp["vec"].to(dev), c["s_txt"], cu, cu, S, S, (cos, sin), c["sa_drop_rate"], c["txt_amp"],
           curve, c["p_remain"])
    yi, yt = db(inp["img"].to(dev), inp["txt"].to(dev), inp["vec"].to(dev), cu, cu, S, S, (cos, sin),
                c["sa_drop_rate"], c["txt_amp"], curve, c["p_remain"])
    for name, got, ref in (("single", y, g["single_out"]), ("double_img", yi, g["double_img"]),
                           ("double_txt", yt, g["double_txt"])):
        got = got.float().cpu().numpy()
        ref = ref.astype(np.float32)
        assert got.shape == ref.shape, (name, got.shape, ref.shape)
        err = np.abs(got - ref)
        bound = 2 * np.exp2(np.floor(np.log2(np.maximum(np.abs(ref), 1e-3))) - 10) + 1.2e-2
        assert (err <= bound).all() and err.mean() <= 1.5e-3, (name, err.max(), err.mean())


def test_jenga_forward_vs_reference_ra_forward(dev):
    """JengaHYVideoDiT.forward against the reference's `ra_forward` (taken out of jenga_hyvideo.py) run over the
    reference's own blocks on CPU in fp16 (tests/golden/make_golden.py gen_hy_forward): a computed step, a skipped step
    replaying the cached residual, another computed step.  Pins the curve gather of tokens and RoPE rows, cu_seqlens,
    the block-call convention, the non_skip_steps list, scatter and unpatchify."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import inputs
    from jenga_amd import dit
    c = inputs.HY_FORWARD
    inp = inputs.hy_forward_inputs()
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hy_forward_case.npz"))
    assert list(g["non_skip_steps"]) == list(dit.NON_SKIP_STEPS)
    dit.HUNYUAN_VIDEO_CONFIG["tiny_fwd"] = dict(mm_double_blocks_depth=c["depth"][0], mm_single_blocks_depth=c["depth"][1],
                                                rope_dim_list=[16, 56, 56], hidden_size=c["hidden"],
                                                heads_num=c["heads"], mlp_width_ratio=c["mlp_ratio"], guidance_embed=True)
    m = dit.JengaHYVideoDiT(config="tiny_fwd", text_states_dim=c["text_dim"], text_states_dim_2=c["text_dim_2"],
                            dtype=torch.float16, device=dev)
    sd = {k_: inputs.hy_param(k_, tuple(v_.shape)) for k_, v_ in m.state_dict().items()}
    m.load_state_dict(sd, strict=True)
    cos, sin = m.set_stage(c["latent"], dev)
    m.enable_skip, m.num_steps, m.start_stage, m.previous_residual = True, 50, False, None
    m.sa_drop_rate, m.text_amp, m.p_remain_rates = c["sa_drop_rate"], c["txt_amp"], c["p_remain"]
    gd = torch.tensor([inp["guidance"]], device=dev)
    for cnt, t in c["steps"]:
        m.cnt = cnt
        y = m(inp["x"].to(dev), torch.tensor([t], device=dev), inp["text"].to(dev), inp["mask"].to(dev),
              inp["text2"].to(dev), cos, sin, gd, return_dict=False)
        assert m.cnt == cnt + 1
        ref = g[f"out_cnt{cnt}"].astype(np.float32)
        got = y.float().cpu().numpy()
        assert got.shape == ref.shape
        err = np.abs(got - ref)
        bound = 2 * np.exp2(np.floor(np.log2(np.maximum(np.abs(ref), 1e-3))) - 10) + 1.5e-2
        assert (err <= bound).all() and err.mean() <= 2e-3, (cnt, err.max(), err.mean())


def test_i2v_single_block_vs_reference_block(dev):
    """MMSingleStreamBlock with token_replace against the reference's HunyuanVideo-I2V block
    (hyvideo_i2v/modules/models_mul.py) run on CPU in fp16: first-frame mask in curve order, four text blocks, the I2V op
    flavour with the Triton kernel under the interpreter (tests/golden/make_golden.py gen_i2v_block)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import inputs
    from jenga_amd import dit
    from jenga_amd.modules.posemb_layers import get_nd_rotary_pos_embed
    c = inputs.I2V_BLOCK
    inp = inputs.i2v_block_inputs()
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "i2v_block_case.npz"))
    h2l = torch.from_numpy(g["hilbert_order"]).to(dev)
    cos, sin = get_nd_rotary_pos_embed([16, 56, 56], list(c["grid"]), theta=256, use_real=True, theta_rescale_factor=1)
    cos, sin = cos.to(dev)[h2l], sin.to(dev)[h2l]
    sb = dit.MMSingleStreamBlock(c["hidden"], c["heads"], c["mlp_ratio"], dtype=torch.float16, device=dev)
    missing, unexpected = sb.load_state_dict(inp["state"], strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    S = inp["S_img"] + c["s_txt"]
    cu = inp["cu"].to(dev)
    y = sb(inp["x"].to(dev), inp["vec"].to(dev), c["s_txt"], cu, cu, S, S, (cos, sin), c["sa_drop_rate"], c["txt_amp"],
           [[None, None, torch.from_numpy(g["neighbors"]).to(dev)]], c["p_remain"], txt_block_num=4,
           token_replace_vec=inp["token_replace_vec"].to(dev),
           first_frame_mask=torch.from_numpy(g["first_frame_mask"]).to(dev))
    got, ref = y.float().cpu().numpy(), g["out"].astype(np.float32)
    assert got.shape == ref.shape
    err = np.abs(got - ref)
    bound = 2 * np.exp2(np.floor(np.log2(np.maximum(np.abs(ref), 1e-3))) - 10) + 1.2e-2
    assert (err <= bound).all() and err.mean() <= 1.5e-3, (err.max(), err.mean())


def test_stage_switch_on_device_vs_reference_composition(golden_dir, dev):
    """Row f-1 on the GPU: prores.switch_stage with device tensors against the fixture generated from the reference
    scheduler (tests/golden/stage_switch_case.npz); fp32 elementwise math + trilinear interpolation: 1e-6."""
    import numpy as np
    from jenga_amd import prores
    g = np.load(os.path.join(golden_dir, "stage_switch_case.npz"))
    shifts = g["shifts"].tolist()
    lat, npred, noise = (torch.from_numpy(g[k]).to(torch.bfloat16).to(dev) for k in ("lat", "npred", "noise"))
    shapes = [tuple(x) for x in g["lat_shapes"].tolist()]
    s = prores.FlowMatchSchedule(50, shift=shifts[0])
    out = prores.switch_stage(s, npred, int(g["split"][0]), lat, shapes[1], shifts[1], noise)
    assert out.device.type == "cuda" and out.dtype == torch.float32
    err = np.abs(out.cpu().numpy() - g["switched"])
    assert err.max() <= 1e-6 * max(1.0, np.abs(g["switched"]).max()), err.max()


def test_wan_turbo_switch_on_device_vs_reference_scheduler(golden_dir, dev):
    """Row f-4 on the GPU: wan_driver.wan_switch_stage with device tensors against the fixture generated from the
    reference FlowUniPCMultistepScheduler (tests/golden/wan_sched_cases.npz); fp32 elementwise math + trilinear
    interpolation: 1e-6 relative."""
    import numpy as np
    from jenga_amd.wan_driver import WanFlowSchedule, wan_switch_stage
    g = np.load(os.path.join(golden_dir, "wan_sched_cases.npz"))
    s = WanFlowSchedule(50, 3.0)
    lat, npred, noise = (torch.from_numpy(g[k]).to(dev) for k in ("lat", "npred", "noise"))
    out = wan_switch_stage(s, npred, 25, lat, (3, 8, 12), noise, 50)
    assert out.device.type == "cuda" and out.dtype == torch.float32
    err = np.abs(out.cpu().numpy() - g["switched"])
    assert err.max() <= 1e-6 * max(1.0, np.abs(g["switched"]).max()), err.max()
    assert np.array_equal(s.sigmas.numpy(), g["sigmas_after"]) and s.shift == 5.0
