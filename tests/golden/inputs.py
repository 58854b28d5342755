"""Seeded input builders shared by make_golden.py (which feeds them to the reference) and the parity tests
(which feed them to the oracle / HIP path).  torch's CPU generator is deterministic for a fixed torch build;
each fixture also stores a sha256 of the inputs so RNG drift is detected instead of silently mis-compared."""
import hashlib

import numpy as np
import torch


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def tensor_sha(t):
    if t.dtype == torch.bfloat16:
        return sha(t.contiguous().view(torch.uint16).numpy())
    return sha(t.contiguous().numpy())


def peaky_qk(gen, B, H, nb_img, nb_all, D, temp):
    """Clustered block means so that the p-rule keeps few blocks (top_k-dominated regime).
    Returns q [B,H,nb_img*128,D], k [B,H,nb_all*128,D] fp32."""
    cent = torch.randn(B, H, nb_all, 1, D, generator=gen) * temp
    k = cent + torch.randn(B, H, nb_all, 128, D, generator=gen)
    pick = torch.randint(0, nb_img, (B, H, nb_img), generator=gen)
    qc = torch.gather(cent[:, :, :nb_img], 2, pick[..., None, None].expand(-1, -1, -1, 1, D))
    q = qc + torch.randn(B, H, nb_img, 128, D, generator=gen)
    return q.reshape(B, H, nb_img * 128, D), k.reshape(B, H, nb_all * 128, D)


# name, flavour, dtype, H, nb_img, text_blocks, top_k, p, peaky_temp, first_frame_blocks
SELECT_SPECS = [
    ("hy_flat_p03", "hy", "bfloat16", 3, 20, 2, 5, 0.3, 0.0, 0),
    ("hy_flat_p09", "hy", "bfloat16", 2, 20, 2, 3, 0.9, 0.0, 0),
    ("hy_peaky_p03", "hy", "bfloat16", 3, 20, 2, 5, 0.3, 1.5, 0),
    ("hy_peaky_p05_fp16", "hy", "float16", 2, 20, 2, 4, 0.5, 1.0, 0),
    ("hy_topk_all", "hy", "bfloat16", 2, 20, 2, 20, 0.3, 0.5, 0),
    ("i2v_text4", "hy", "bfloat16", 2, 20, 4, 6, 0.3, 0.7, 0),
    ("wan_ff2", "wan", "bfloat16", 2, 20, 0, 6, 0.8, 0.7, 2),
    ("wan_flat", "wan", "bfloat16", 2, 20, 0, 4, 0.5, 0.0, 3),
]
SELECT_GRID = (4, 8, 80)  # 2560 tokens = 20 blocks of 128


def select_inputs(index):
    """-> q [1,H,nb_img*128,128], k [1,H,nb_all*128,128] in the case dtype."""
    name, flav, dt, H, nb_img, tb, top_k, p, temp, ffb = SELECT_SPECS[index]
    dt = getattr(torch, dt)
    gen = torch.Generator().manual_seed(1000 + index)
    nb_all = nb_img + tb
    if temp > 0:
        q, k = peaky_qk(gen, 1, H, nb_img, nb_all, 128, temp)
    else:
        q = torch.randn(1, H, nb_img * 128, 128, generator=gen)
        k = torch.randn(1, H, nb_all * 128, 128, generator=gen)
    return q.to(dt), k.to(dt)


# H, nb_img, text_blocks, valid text tokens, text_amp, seed
KERNEL_SPECS = [(2, 4, 2, 70, 0.0, 7), (2, 5, 2, 256, 0.431, 8), (1, 3, 1, 1, 0.25, 9)]


NARROW_HEAD_DIMS = (64, 32, 16)        # the other head dims the Triton kernel accepts (:155)


def narrow_kernel_inputs(D, dtype=torch.bfloat16):
    """Kernel case with D-channel heads: -> q [1,2,3*128,D], k, v [1,2,4*128,D], mask, seqlen, text_amp."""
    H, nb_img, tb = 2, 3, 1
    gen = torch.Generator().manual_seed(100 + D)
    S = (nb_img + tb) * 128
    amp = 1.6 * (128.0 / D) ** 0.5      # same spread of the scaled scores as the 128-channel cases
    q = (torch.randn(1, H, nb_img * 128, D, generator=gen) * amp).to(dtype)
    k = (torch.randn(1, H, S, D, generator=gen) * 1.2).to(dtype)
    v = torch.randn(1, H, S, D, generator=gen).to(dtype)
    mask = torch.rand(1, H, nb_img, nb_img + tb, generator=gen) < 0.5
    mask[..., nb_img:] = True
    for i in range(nb_img):
        mask[:, :, i, i] = True
    return q, k, v, mask, nb_img * 128 + 77, 0.3


def kernel_inputs(index, dtype=torch.float16):
    """-> q [1,H,nb_img*128,128], k, v [1,H,S,128] fp16 (or the same draws rounded to `dtype`), mask bool
    [1,H,nb_img,nb_all], seqlen, text_amp."""
    H, nb_img, tb, seqlen_txt, amp, seed = KERNEL_SPECS[index]
    gen = torch.Generator().manual_seed(seed)
    S = (nb_img + tb) * 128
    q = (torch.randn(1, H, nb_img * 128, 128, generator=gen) * 1.2).to(dtype)
    k = (torch.randn(1, H, S, 128, generator=gen) * 1.2).to(dtype)
    v = torch.randn(1, H, S, 128, generator=gen).to(dtype)
    mask = torch.rand(1, H, nb_img, nb_img + tb, generator=gen) < 0.5
    mask[..., nb_img:] = True
    for i in range(nb_img):
        mask[:, :, i, i] = True
    return q, k, v, mask, nb_img * 128 + seqlen_txt, amp


OP_SPEC = dict(H=2, nb_img=6, text_blocks=2, top_k=2, p=0.3, text_amp=0.3, valid_text=100, grid=(2, 8, 48), seed=21)


def op_inputs():
    """Whole-op case: q,k,v [1,S,H,128] fp16 (peaky), cu_seqlens int32 [3]."""
    s = OP_SPEC
    gen = torch.Generator().manual_seed(s["seed"])
    nb = s["nb_img"] + s["text_blocks"]
    S = nb * 128
    q, k = peaky_qk(gen, 1, s["H"], nb, nb, 128, 0.8)
    q = q.transpose(1, 2).half().contiguous()
    k = k.transpose(1, 2).half().contiguous()
    v = torch.randn(1, S, s["H"], 128, generator=gen).half()
    cu = torch.tensor([0, s["nb_img"] * 128 + s["valid_text"], S], dtype=torch.int32)
    return q, k, v, cu


# ---- Wan attention block (wan/modules/model_mul.py:252-346) -------------------------------------------------------
WAN_BLOCK = dict(dim=256, ffn_dim=512, num_heads=2, grid=(2, 8, 8), ctx_len=96, eps=1e-6)


def wan_block_inputs():
    """Seeded state dict (reference parameter names), x fp32 [1,128,256] (bf16-representable: the first block sees
    the 16-bit patch embedding), e fp32 [1,6,256], context bf16 [1,96,256], freq_remap (a permutation of the
    tokens).  Linear weights are bf16-representable so that autocast's on-the-fly cast is exact."""
    c = WAN_BLOCK
    dim, ffn = c["dim"], c["ffn_dim"]
    gen = torch.Generator().manual_seed(4242)
    bf = lambda t: t.to(torch.bfloat16).float()
    sd = {}
    for pre in ("self_attn", "cross_attn"):
        for n in ("q", "k", "v", "o"):
            sd[f"{pre}.{n}.weight"] = bf(torch.randn(dim, dim, generator=gen) * 0.06)
            sd[f"{pre}.{n}.bias"] = bf(torch.randn(dim, generator=gen) * 0.05)
        sd[f"{pre}.norm_q.weight"] = 1 + 0.1 * torch.randn(dim, generator=gen)
        sd[f"{pre}.norm_k.weight"] = 1 + 0.1 * torch.randn(dim, generator=gen)
    sd["norm3.weight"] = 1 + 0.1 * torch.randn(dim, generator=gen)
    sd["norm3.bias"] = 0.1 * torch.randn(dim, generator=gen)
    sd["ffn.0.weight"] = bf(torch.randn(ffn, dim, generator=gen) * 0.06)
    sd["ffn.0.bias"] = bf(torch.randn(ffn, generator=gen) * 0.05)
    sd["ffn.2.weight"] = bf(torch.randn(dim, ffn, generator=gen) * 0.05)
    sd["ffn.2.bias"] = bf(torch.randn(dim, generator=gen) * 0.05)
    sd["modulation"] = torch.randn(1, 6, dim, generator=gen) / dim ** 0.5
    f, h, w = c["grid"]
    L = f * h * w
    x = bf(torch.randn(1, L, dim, generator=gen) * 1.3)
    e = torch.randn(1, 6, dim, generator=gen) * 0.3
    ctx = (torch.randn(1, c["ctx_len"], dim, generator=gen)).to(torch.bfloat16)
    remap = torch.randperm(L, generator=gen)
    return dict(state=sd, x=x, e=e, context=ctx, remap=remap)


# ---- HunyuanVideo DiT blocks (hyvideo/modules/models_mul_block_gc_ha_multigpu.py:41-316, 318-500) -----------------
HY_BLOCK = dict(hidden=256, heads=2, mlp_ratio=4, grid=(2, 8, 32), s_txt=256, valid_txt=70, sa_drop_rate=0.5,
                txt_amp=0.3, p_remain=0.3, dtype="float16")


def _lin(gen, out_f, in_f, std):
    return torch.randn(out_f, in_f, generator=gen) * std, torch.randn(out_f, generator=gen) * 0.02


def hy_block_inputs():
    """Seeded fp16 state dicts with the reference's parameter names for one MMSingleStreamBlock and one
    MMDoubleStreamBlock, their inputs and cu_seqlens.  fp16 because the reference's Triton kernel only runs in fp16
    under the CPU interpreter."""
    c = HY_BLOCK
    C, H = c["hidden"], c["heads"]
    M = C * c["mlp_ratio"]
    gen = torch.Generator().manual_seed(777)
    h = lambda t: t.to(torch.float16)
    single = {}
    w, b = _lin(gen, 3 * C + M, C, 0.06); single["linear1.weight"], single["linear1.bias"] = h(w), h(b)
    w, b = _lin(gen, C, C + M, 0.05); single["linear2.weight"], single["linear2.bias"] = h(w), h(b)
    single["q_norm.weight"] = h(1 + 0.1 * torch.randn(128, generator=gen))
    single["k_norm.weight"] = h(1 + 0.1 * torch.randn(128, generator=gen))
    w, b = _lin(gen, 3 * C, C, 0.04); single["modulation.linear.weight"], single["modulation.linear.bias"] = h(w), h(b)
    double = {}
    for s_ in ("img", "txt"):
        w, b = _lin(gen, 6 * C, C, 0.04); double[f"{s_}_mod.linear.weight"], double[f"{s_}_mod.linear.bias"] = h(w), h(b)
        w, b = _lin(gen, 3 * C, C, 0.06); double[f"{s_}_attn_qkv.weight"], double[f"{s_}_attn_qkv.bias"] = h(w), h(b)
        double[f"{s_}_attn_q_norm.weight"] = h(1 + 0.1 * torch.randn(128, generator=gen))
        double[f"{s_}_attn_k_norm.weight"] = h(1 + 0.1 * torch.randn(128, generator=gen))
        w, b = _lin(gen, C, C, 0.06); double[f"{s_}_attn_proj.weight"], double[f"{s_}_attn_proj.bias"] = h(w), h(b)
        w, b = _lin(gen, M, C, 0.06); double[f"{s_}_mlp.fc1.weight"], double[f"{s_}_mlp.fc1.bias"] = h(w), h(b)
        w, b = _lin(gen, C, M, 0.04); double[f"{s_}_mlp.fc2.weight"], double[f"{s_}_mlp.fc2.bias"] = h(w), h(b)
    f_, hh, ww = c["grid"]
    S_img = f_ * hh * ww
    x = h(torch.randn(1, S_img + c["s_txt"], C, generator=gen))
    img = h(torch.randn(1, S_img, C, generator=gen))
    txt = h(torch.randn(1, c["s_txt"], C, generator=gen))
    vec = h(torch.randn(1, C, generator=gen))
    cu = torch.tensor([0, S_img + c["valid_txt"], S_img + c["s_txt"]], dtype=torch.int32)
    return dict(single=single, double=double, x=x, img=img, txt=txt, vec=vec, cu=cu, S_img=S_img)


# ---- Wan2.1 model forward (wan/modules/model_mul.py WanModel + jenga_wan.py teacache_forward) -----------------------
WAN_MODEL = dict(dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64, text_len=32, in_dim=16, out_dim=16,
                 freq_dim=256, latent=(3, 8, 16), steps=7, thresh=0.15, ctx_len=(20, 7))


def wan_param(key, shape, fan_in_gain=None):
    """Deterministic tensor for the parameter `key` (reference names), independent of iteration order: linear weights
    and biases are bf16-representable (autocast's cast of them is then exact), norms / modulation stay fp32.
    fan_in_gain: weights get std = gain / sqrt(fan_in) instead of 0.05 (full-size models)."""
    import zlib
    gen = torch.Generator().manual_seed(zlib.crc32(key.encode()) & 0x7FFFFFFF)
    t = torch.randn(*shape, generator=gen)
    if fan_in_gain is not None and key.endswith("weight") and len(shape) >= 2 and "norm" not in key:
        fan_in = 1
        for d in shape[1:]:
            fan_in *= d
        w = t * (fan_in_gain / fan_in ** 0.5)
        return w if key.startswith(("time_embedding", "time_projection", "head.head")) else w.to(torch.bfloat16).float()
    if key.endswith("modulation"):
        return t / shape[-1] ** 0.5
    if "norm" in key:
        return (1 + 0.1 * t) if key.endswith("weight") else 0.1 * t
    if key.startswith(("time_embedding", "time_projection", "head.head")):      # run in fp32 (autocast(float32))
        return t * (0.05 if key.endswith("weight") else 0.02)
    return (t * (0.05 if key.endswith("weight") else 0.02)).to(torch.bfloat16).float()


def wan_model_inputs():
    c = WAN_MODEL
    gen = torch.Generator().manual_seed(99)
    x = torch.randn(c["in_dim"], *c["latent"], generator=gen)
    ctx = [torch.randn(n, c["text_dim"], generator=gen) for n in c["ctx_len"]]      # cond / uncond prompts
    ts = [900.0, 899.95, 899.9, 899.8, 899.5, 899.0, 898.0][: c["steps"]]
    return dict(x=x, context=ctx, timesteps=ts)


# ---- HunyuanVideo Jenga forward (jenga_hyvideo.py ra_forward) -------------------------------------------------------
HY_FORWARD = dict(hidden=256, heads=2, mlp_ratio=4, depth=(1, 1), latent=(4, 16, 32), text_len=256, text_dim=64,
                  text_dim_2=32, valid_txt=70, sa_drop_rate=0.5, txt_amp=0.2, p_remain=0.3, dtype="float16",
                  steps=((0, 900.0), (5, 820.0), (7, 700.0)))       # (cnt, timestep): computed, skipped, computed


def hy_param(key, shape):
    """Deterministic fp16-representable tensor for parameter `key` of the tiny DiT (names shared by the reference
    blocks and jenga_amd.dit)."""
    import zlib
    gen = torch.Generator().manual_seed(zlib.crc32(("hy:" + key).encode()) & 0x7FFFFFFF)
    t = torch.randn(*shape, generator=gen)
    if "_norm.weight" in key or key.endswith("norm.weight"):
        return (1 + 0.1 * t).to(torch.float16)
    return (t * (0.05 if len(shape) >= 2 else 0.02)).to(torch.float16)


def hy_forward_inputs():
    c = HY_FORWARD
    gen = torch.Generator().manual_seed(2024)
    x = torch.randn(1, 16, *c["latent"], generator=gen).to(torch.float16)
    text = torch.randn(1, c["text_len"], c["text_dim"], generator=gen).to(torch.float16)
    text2 = torch.randn(1, c["text_dim_2"], generator=gen).to(torch.float16)
    mask = torch.zeros(1, c["text_len"], dtype=torch.int64)
    mask[:, : c["valid_txt"]] = 1
    return dict(x=x, text=text, text2=text2, mask=mask, guidance=6000.0)


# ---- HunyuanVideo-I2V single-stream block with token_replace (hyvideo_i2v/modules/models_mul.py:393-506) -----------
I2V_BLOCK = dict(hidden=256, heads=2, mlp_ratio=4, grid=(2, 8, 32), s_txt=512, valid_txt=300, sa_drop_rate=0.5,
                 txt_amp=0.3, p_remain=0.3, dtype="float16")


def i2v_block_inputs():
    c = I2V_BLOCK
    C = c["hidden"]
    M = C * c["mlp_ratio"]
    keys = {"linear1.weight": (3 * C + M, C), "linear1.bias": (3 * C + M,), "linear2.weight": (C, C + M),
            "linear2.bias": (C,), "q_norm.weight": (128,), "k_norm.weight": (128,),
            "modulation.linear.weight": (3 * C, C), "modulation.linear.bias": (3 * C,)}
    sd = {k_: hy_param("i2v." + k_, shp) for k_, shp in keys.items()}
    gen = torch.Generator().manual_seed(31337)
    f_, hh, ww = c["grid"]
    S_img = f_ * hh * ww
    x = torch.randn(1, S_img + c["s_txt"], C, generator=gen).to(torch.float16)
    vec = torch.randn(1, C, generator=gen).to(torch.float16)
    trv = torch.randn(1, C, generator=gen).to(torch.float16)
    cu = torch.tensor([0, S_img + c["valid_txt"], S_img + c["s_txt"]], dtype=torch.int32)
    return dict(state=sd, x=x, vec=vec, token_replace_vec=trv, cu=cu, S_img=S_img)


# ---- BASELINE.json configs[0]: Wan2.1-1.3B T2V 256x256x17f, dense, the reference's CPU-eager case -----------------
WAN_1P3B = dict(dim=1536, ffn_dim=8960, num_heads=12, num_layers=30, text_dim=4096, text_len=512, in_dim=16, out_dim=16,
                freq_dim=256, latent=(5, 32, 32), ctx_len=40, timestep=999.0, gain=0.8)


def wan_1p3b_inputs():
    c = WAN_1P3B
    gen = torch.Generator().manual_seed(1313)
    x = torch.randn(c["in_dim"], *c["latent"], generator=gen)
    ctx = torch.randn(c["ctx_len"], c["text_dim"], generator=gen)
    return dict(x=x, context=ctx)
