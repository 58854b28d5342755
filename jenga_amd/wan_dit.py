"""Wan2.1 DiT around the AttenCarve self-attention: counterpart of wan/modules/model_mul.py
   WanT2VCrossAttention :183-205, WanAttentionBlock :252-346, Head :349-378, WanModel :392-520
and of the Jenga forward that replaces WanModel.forward (jenga_wan.py:503-664, teacache_forward).

What runs where: the residual stream is fp32 (the reference evaluates `x + y * e` under autocast(float32)); the
LayerNorm + modulation that feed the GEMMs and the gated residual adds are the two fused HIP kernels
jenga_wan_ln_modulate / jenga_wan_gate_residual; self-attention is jenga_amd.modules.wan.WanSelfAttention (full-width
RMSNorm, fp64 complex RoPE, block selection + block-sparse attention kernels); GELU is jenga_gelu_tanh; linear layers
are hipBLASLt (jenga_linear where an epilogue rides along, torch otherwise); the text cross-attention (512 keys, 0.4 % of
the FLOPs) runs on the LP attention kernel's dense mode (jenga_cross_attn_fwd) -- no library attention kernel anywhere.  Weights are random-initialised here (no checkpoints in this environment);
state-dict keys follow the reference so that a Wan checkpoint loads with `patch_embedding.weight` flattened."""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _capi
from .modules.wan import WanRMSNorm, WanSelfAttention, wan_freqs
from .wan_driver import TeaCache, teacache_forward

WAN_CONFIGS = {   # wan/configs/wan_t2v_14B.py, wan_t2v_1_3B.py
    "t2v-14B": dict(dim=5120, ffn_dim=13824, num_heads=40, num_layers=40),
    "t2v-1.3B": dict(dim=1536, ffn_dim=8960, num_heads=12, num_layers=30),
}


def sinusoidal_embedding_1d(dim, position):
    """model_mul.py:16-27: float64 [len(position), dim] = [cos | sin]."""
    half = dim // 2
    position = position.to(torch.float64)
    sinusoid = torch.outer(position, torch.pow(10000, -torch.arange(half, device=position.device).to(position).div(half)))
    return torch.cat([torch.cos(sinusoid), torch.sin(sinusoid)], dim=1)


class WanLayerNormParams(nn.Module):
    """Holder with nn.LayerNorm's parameter names (norm3 has an affine; norm1 / norm2 do not)."""

    def __init__(self, dim, eps, elementwise_affine, device=None):
        super().__init__()
        self.eps = eps
        if elementwise_affine:
            self.weight = nn.Parameter(torch.ones(dim, device=device))
            self.bias = nn.Parameter(torch.zeros(dim, device=device))
        else:
            self.weight = self.bias = None


class WanT2VCrossAttention(nn.Module):
    def __init__(self, dim, num_heads, eps=1e-6, dtype=None, device=None):
        super().__init__()
        self.num_heads, self.head_dim = num_heads, dim // num_heads
        fk = dict(dtype=dtype, device=device)
        self.q, self.k, self.v, self.o = (nn.Linear(dim, dim, **fk) for _ in range(4))
        self.norm_q = WanRMSNorm(dim, eps=eps).to(device)
        self.norm_k = WanRMSNorm(dim, eps=eps).to(device)

    @torch.no_grad()
    def forward(self, x, context, context_lens=None, project=True):
        if context_lens is not None:
            raise ValueError("jenga_amd Wan cross-attention: context_lens must be None (the Jenga driver passes None)")
        b, n, d = x.shape[0], self.num_heads, self.head_dim
        L, Lc = x.shape[1], context.shape[1]
        if b != 1 or d != 128 or Lc == 0 or context.dtype not in (torch.bfloat16, torch.float16):
            raise ValueError("jenga_amd Wan cross-attention: batch 1, head_dim 128, a non-empty 16-bit context")
        # q, k, v live in buffers padded to whole 128-row blocks (query rows are independent; keys >= Lc are masked by the
        # kernel -- the reference's context is always text_len = 512 = four whole blocks)
        Lp, Lcp = (L + 127) // 128 * 128, (Lc + 127) // 128 * 128
        dt = context.dtype
        kv = torch.empty((2, b, Lcp, n * d), dtype=dt, device=x.device)
        if Lcp > Lc:
            kv[:, :, Lc:].zero_()
        kv[1, :, :Lc] = self.v(context)
        # WanRMSNorm with its fp32 weight returns fp32; flash_attention's half() rounds q, k to the 16-bit dtype: norm
        # (fp32 weight) + the cast in one pass
        q = torch.empty((b, Lp, n * d), dtype=dt, device=x.device)
        if dt == torch.bfloat16:
            _capi.wan_norm_rope(self.q(x), self.norm_q.weight, None, None, 0, self.norm_q.eps, out=q.view(Lp, n * d))
            _capi.wan_norm_rope(self.k(context), self.norm_k.weight, None, None, 0, self.norm_k.eps,
                                out=kv[0].view(Lcp, n * d))
        else:
            q[:, :L] = self.norm_q(self.q(x)).to(dt)
            kv[0, :, :Lc] = self.norm_k(self.k(context)).to(dt)
        if Lp > L:
            q[:, L:].zero_()
        # softmax scale d^-0.5, no mask beyond the context length (flash_attention(k_lens=None)): the LP attention kernel in
        # its dense mode, every query block against the context blocks (jenga_cross_attn_fwd)
        o = _capi.cross_attn_fwd(q.view(b, Lp, n, d), kv[0].view(b, Lcp, n, d), kv[1].view(b, Lcp, n, d), kv_len=Lc)
        return self.o(o[:, :L].flatten(2)) if project else o[:, :L].flatten(2)


# Round 6: gate + residual of the Wan blocks in the GEMM epilogue (the HunyuanVideo blocks have had it since round 3).  One
# rounding (to fp32) where the reference rounds the linear layer's output to 16 bits first; JENGA_WAN_FUSE_GATE=0 is the
# configuration that is bit-comparable with the reference blocks (tests/test_gpu_wan_dit.py runs both).
WAN_FUSE_GATE = os.environ.get("JENGA_WAN_FUSE_GATE", "1") != "0"


def _linear_into_stream(lin, a, gate, x, out=None):
    """x (fp32 residual stream) + gate * lin(a): the gate is hipBLASLt's alpha vector, x its C matrix, the gated bias stays fp32."""
    g = None if gate is None else gate.reshape(-1).float()
    bias = None
    if lin.bias is not None:
        bias = lin.bias.float() * g if g is not None else lin.bias.float()
    return _capi.linear(a, lin.weight, bias, gate=g, res=x, out=out)


class WanAttentionBlock(nn.Module):
    """forward(x fp32 [1,L,C], e fp32 [1,6,C], ...) with the reference's argument list (model_mul.py:300-346)."""

    def __init__(self, cross_attn_type, dim, ffn_dim, num_heads, window_size=(-1, -1), qk_norm=True,
                 cross_attn_norm=False, eps=1e-6, index=0, num_layers=0, dtype=torch.bfloat16, device=None):
        super().__init__()
        if cross_attn_type != "t2v_cross_attn":
            raise ValueError("only the t2v cross-attention is built")
        self.dim, self.ffn_dim, self.num_heads, self.eps = dim, ffn_dim, num_heads, eps
        fk = dict(dtype=dtype, device=device)
        self.norm1 = WanLayerNormParams(dim, eps, False)
        self.self_attn = WanSelfAttention(dim, num_heads, window_size, qk_norm, eps, index=index,
                                          num_layers=num_layers, **fk)
        self.norm3 = WanLayerNormParams(dim, eps, cross_attn_norm, device=device)
        self.cross_attn_norm = cross_attn_norm
        self.cross_attn = WanT2VCrossAttention(dim, num_heads, eps, **fk)
        self.norm2 = WanLayerNormParams(dim, eps, False)
        self.ffn = nn.Sequential(nn.Linear(dim, ffn_dim, **fk), nn.GELU(approximate="tanh"),
                                 nn.Linear(ffn_dim, dim, **fk))
        self.modulation = nn.Parameter(torch.randn(1, 6, dim, device=device) / dim ** 0.5)

    def ffn_hidden(self, h):
        """ffn[0] + tanh-GELU: the activation rides in the GEMM's epilogue (jenga_linear) -- one pass less
        over the [L, ffn_dim] activations (4.2 GB per layer at the 14B 720p shape)."""
        return _capi.linear(h, self.ffn[0].weight, self.ffn[0].bias, act=_capi.ACT_GELU_TANH)

    @torch.no_grad()
    def forward(self, x, e, seq_lens, grid_sizes, freqs, context, context_lens, sa_drop_rate=0.0, freq_remap=None,
                block_neighbor_list=None, p_remain_rates=0.0, x_was_16bit=False):
        if x.dtype != torch.float32 or e.dtype != torch.float32 or x.shape[0] != 1:
            raise ValueError("Wan block: x and e are float32 and the batch is 1")
        em = self.modulation.float() + e                                         # [1,6,C]
        if WAN_FUSE_GATE and x.is_cuda:
            # the three `x = x + y [* e]` updates of the fp32 residual stream (model_mul.py:334-341) in the epilogue of the
            # GEMM that produces y (jenga_linear, JENGA_OUT_F32): no separate 3.9 GB gate + residual pass per update
            h = _capi.wan_ln_modulate(x, shift=em[:, 0], scale=em[:, 1], eps=self.eps, round_ln=x_was_16bit)
            a = self.self_attn(h, seq_lens, grid_sizes, freqs, sa_drop_rate=sa_drop_rate, freq_remap=freq_remap,
                               block_neighbor_list=block_neighbor_list, p_remain_rates=p_remain_rates, project=False)
            x = _linear_into_stream(self.self_attn.o, a, em[:, 2], x)              # new tensor: callers keep their input
            h = _capi.wan_ln_modulate(x, weight=self.norm3.weight, bias=self.norm3.bias, eps=self.eps) \
                if self.cross_attn_norm else x.to(context.dtype)
            _linear_into_stream(self.cross_attn.o, self.cross_attn(h, context, context_lens, project=False), None, x, out=x)
            h = _capi.wan_ln_modulate(x, shift=em[:, 3], scale=em[:, 4], eps=self.eps)
            _linear_into_stream(self.ffn[2], self.ffn_hidden(h), em[:, 5], x, out=x)
            return x
        # self-attention: y = attn(norm1(x) * (1 + e1) + e0);  x = x + y * e2
        h = _capi.wan_ln_modulate(x, shift=em[:, 0], scale=em[:, 1], eps=self.eps, round_ln=x_was_16bit)
        y = self.self_attn(h, seq_lens, grid_sizes, freqs, sa_drop_rate=sa_drop_rate, freq_remap=freq_remap,
                           block_neighbor_list=block_neighbor_list, p_remain_rates=p_remain_rates)
        x = _capi.wan_gate_residual(x, y, em[:, 2])                             # new tensor: callers keep their input
        # cross-attention: x = x + cross_attn(norm3(x), context)
        h = _capi.wan_ln_modulate(x, weight=self.norm3.weight, bias=self.norm3.bias, eps=self.eps) \
            if self.cross_attn_norm else x.to(context.dtype)
        _capi.wan_gate_residual(x, self.cross_attn(h, context, context_lens), None, out=x)
        # ffn: y = ffn(norm2(x) * (1 + e4) + e3);  x = x + y * e5
        h = _capi.wan_ln_modulate(x, shift=em[:, 3], scale=em[:, 4], eps=self.eps)
        y = self.ffn[2](self.ffn_hidden(h))
        _capi.wan_gate_residual(x, y, em[:, 5], out=x)
        return x


class Head(nn.Module):
    def __init__(self, dim, out_dim, patch_size, eps=1e-6, device=None):
        super().__init__()
        self.dim, self.out_dim, self.patch_size, self.eps = dim, out_dim, patch_size, eps
        self.head = nn.Linear(dim, math.prod(patch_size) * out_dim, device=device)     # fp32: runs under autocast(fp32)
        self.modulation = nn.Parameter(torch.randn(1, 2, dim, device=device) / dim ** 0.5)

    @torch.no_grad()
    def forward(self, x, e):
        em = (self.modulation.float() + e.unsqueeze(1)).chunk(2, dim=1)
        return self.head(F.layer_norm(x.float(), (self.dim,), eps=self.eps) * (1 + em[1]) + em[0])


class WanDiT(nn.Module):
    """WanModel (t2v) with the Jenga forward.  set_curve() installs the (sliced) Gilbert tables of the current
    resolution the way jenga_wan.py:1026-1034, 1079-1081 does."""

    def __init__(self, model_type="t2v", patch_size=(1, 2, 2), text_len=512, in_dim=16, dim=2048, ffn_dim=8192,
                 freq_dim=256, text_dim=4096, out_dim=16, num_heads=16, num_layers=32, window_size=(-1, -1),
                 qk_norm=True, cross_attn_norm=True, eps=1e-6, dtype=torch.bfloat16, device=None):
        super().__init__()
        if model_type != "t2v":
            raise ValueError("only t2v is built")
        self.patch_size, self.text_len, self.in_dim, self.dim = patch_size, text_len, in_dim, dim
        self.freq_dim, self.out_dim, self.num_heads, self.num_layers = freq_dim, out_dim, num_heads, num_layers
        fk = dict(dtype=dtype, device=device)
        # Conv3d(in_dim, dim, kernel = stride = patch) == a linear map of each patch (c, kt, kh, kw)-flattened
        self.patch_embedding = nn.Linear(in_dim * math.prod(patch_size), dim, **fk)
        self.text_embedding = nn.Sequential(nn.Linear(text_dim, dim, **fk), nn.GELU(approximate="tanh"),
                                            nn.Linear(dim, dim, **fk))
        self.time_embedding = nn.Sequential(nn.Linear(freq_dim, dim, device=device), nn.SiLU(),
                                            nn.Linear(dim, dim, device=device))              # fp32 (autocast(fp32))
        self.time_projection = nn.Sequential(nn.SiLU(), nn.Linear(dim, dim * 6, device=device))
        self.blocks = nn.ModuleList([
            WanAttentionBlock("t2v_cross_attn", dim, ffn_dim, num_heads, window_size, qk_norm, cross_attn_norm, eps,
                              index=i, num_layers=num_layers, **fk) for i in range(num_layers)])
        self.head = Head(dim, out_dim, patch_size, eps, device=device)
        self.freqs = wan_freqs(dim // num_heads)
        self.p_remain_rates = 0.8
        self.hilbert_order = self.linear_to_hilbert = self.block_neighbor_list = None
        self.tea = None
        self.last_computed = None      # whether the last forward ran the blocks (False: TeaCache replayed a residual)

    def set_curve(self, linear_to_hilbert, hilbert_order, block_neighbor_list):
        self.linear_to_hilbert, self.hilbert_order, self.block_neighbor_list = (linear_to_hilbert, hilbert_order,
                                                                                 block_neighbor_list)

    def enable_teacache(self, num_steps, thresh, task, use_ret_steps=False, enable=True):
        self.tea = TeaCache(num_steps, thresh, task, use_ret_steps, enable)

    def patchify(self, u):
        """[C, F, H, W] -> tokens [F*h*w, C*pt*ph*pw] in Conv3d's (c, kt, kh, kw) weight order, grid (F/pt, H/ph, W/pw)."""
        C, Fr, H, W = u.shape
        pt, ph, pw = self.patch_size
        f, h, w = Fr // pt, H // ph, W // pw
        t = u.view(C, f, pt, h, ph, w, pw).permute(1, 3, 5, 0, 2, 4, 6).reshape(f * h * w, C * pt * ph * pw)
        return t, (f, h, w)

    def unpatchify(self, x, grid):
        """[f*h*w, out_dim*pt*ph*pw] -> [out_dim, F, H, W] (model_mul.py:596-617)."""
        c = self.out_dim
        f, h, w = grid
        u = x[: f * h * w].view(f, h, w, *self.patch_size, c)
        u = torch.einsum("fhwpqrc->cfphqwr", u)
        return u.reshape(c, f * self.patch_size[0], h * self.patch_size[1], w * self.patch_size[2])

    @torch.no_grad()
    def forward(self, x, t, context, seq_len, sa_drop_rate=0.0):
        """x: list with one [C_in, F, H, W] latent; t [1]; context: list with one [L, text_dim]; returns [tensor fp32]."""
        if len(x) != 1 or len(context) != 1:
            raise ValueError("jenga_amd WanDiT: batch of one (the driver calls cond / uncond separately)")
        if self.hilbert_order is None or self.tea is None:
            raise RuntimeError("call set_curve() and enable_teacache() first")
        dev = self.patch_embedding.weight.device
        wd = self.patch_embedding.weight.dtype
        tok, grid = self.patchify(x[0].to(dev))
        tokens = self.patch_embedding(tok.to(wd)).unsqueeze(0)                      # 16-bit, like the Conv3d output
        grid_sizes = torch.tensor([list(grid)], dtype=torch.long)
        seq_lens = torch.tensor([tokens.shape[1]], dtype=torch.long)
        assert int(seq_lens.max()) <= seq_len
        e = self.time_embedding(sinusoidal_embedding_1d(self.freq_dim, t.to(dev)).float())
        e0 = self.time_projection(e).unflatten(1, (6, self.dim))
        ctx = context[0].to(dev)
        ctx = torch.cat([ctx, ctx.new_zeros(self.text_len - ctx.shape[0], ctx.shape[1])]).unsqueeze(0)
        ctx = self.text_embedding(ctx.to(wd))
        first = {"flag": True}

        def run_block(blk):
            def call(xx, **kw):
                out = blk(xx, x_was_16bit=first["flag"], **kw)
                first["flag"] = False
                return out
            return call

        out, self.last_computed = teacache_forward(tokens.float(), e, e0, [run_block(b) for b in self.blocks], self.tea,
                                  self.hilbert_order, self.linear_to_hilbert, seq_len=seq_len, e=e0, seq_lens=seq_lens,
                                  grid_sizes=grid_sizes, freqs=self.freqs, context=ctx, context_lens=None,
                                  sa_drop_rate=sa_drop_rate, freq_remap=self.hilbert_order,
                                  block_neighbor_list=self.block_neighbor_list, p_remain_rates=self.p_remain_rates)
        y = self.head(out, e)
        return [self.unpatchify(y[0], grid).float()]
