"""Wan2.1 flavour of the AttenCarve caller, counterpart of wan/modules/model_mul.py:
   rope_params :29-37, rope_apply :40-71, WanRMSNorm :74-90, WanSelfAttention :108-180.

Numerics on the GPU: jenga_rmsnorm_rows (full-width RMSNorm), jenga_rope_complex (float64 complex rotation),
jenga_amd.modules.attention_block_sparse.block_sparse_attention_wan (selection + block-sparse attention).
The linear layers are plain torch (hipBLASLt)."""
import math

import torch
import torch.nn as nn

from .. import _capi
from . import attention_block_sparse as _op


def rope_params(max_seq_len, dim, theta=10000):
    """complex128 [max_seq_len, dim/2] -- same torch calls as the reference (:29-37)."""
    assert dim % 2 == 0
    freqs = torch.outer(torch.arange(max_seq_len),
                        1.0 / torch.pow(theta, torch.arange(0, dim, 2).to(torch.float64).div(dim)))
    return torch.polar(torch.ones_like(freqs), freqs)


def wan_freqs(head_dim=128):
    """The model-level table (model_mul.py:502-507): three axis tables concatenated to [1024, head_dim/2]."""
    d = head_dim
    return torch.cat([rope_params(1024, d - 4 * (d // 6)), rope_params(1024, 2 * (d // 6)),
                      rope_params(1024, 2 * (d // 6))], dim=1)


def expand_freqs(freqs, grid, freq_remap=None):
    """Per-token multipliers for an (f, h, w) grid, split [c-2(c//3), c//3, c//3] and broadcast exactly as rope_apply
    does (:45-65) -> (cos, sin) float64 [f*h*w, c].  Static per resolution: build once, keep on the device."""
    f, h, w = grid
    c = freqs.shape[1]
    parts = freqs.split([c - 2 * (c // 3), c // 3, c // 3], dim=1)
    fi = torch.cat([parts[0][:f].view(f, 1, 1, -1).expand(f, h, w, -1),
                    parts[1][:h].view(1, h, 1, -1).expand(f, h, w, -1),
                    parts[2][:w].view(1, 1, w, -1).expand(f, h, w, -1)], dim=-1).reshape(f * h * w, c)
    if freq_remap is not None:
        fi = fi[freq_remap.to(fi.device)]
    return fi.real.contiguous(), fi.imag.contiguous()


def rope_tables(grid_sizes, freqs, freq_remap, device, _cache={}):
    """-> (cos, sin float64 [f*h*w, 64] on the device, f*h*w) for the batch's common (f, h, w) grid; built once per
    resolution (static geometry) and kept."""
    grids = grid_sizes.tolist() if torch.is_tensor(grid_sizes) else list(grid_sizes)
    if len({tuple(g) for g in grids}) != 1:
        raise ValueError("jenga_amd rope_apply: all samples of the batch must share one (f,h,w) grid")
    f, h, w = grids[0]
    key = (f, h, w, device, None if freq_remap is None else freq_remap.data_ptr(), freqs.data_ptr())
    if key not in _cache:
        cos, sin = expand_freqs(freqs, (f, h, w), freq_remap)
        _cache.clear()
        _cache[key] = (cos.to(device), sin.to(device))
    cos, sin = _cache[key]
    return cos, sin, f * h * w


def rope_apply(x, grid_sizes, freqs, freq_remap=None, out_dtype=torch.float32):
    """x [B,S,N,128]; grid_sizes [B,3]; freqs complex128 [1024,64] -> float32 [B,S,N,128] (the reference returns
    `.float()`); out_dtype=torch.bfloat16 fuses the cast the Wan attention op applies next."""
    cos, sin, n_rope = rope_tables(grid_sizes, freqs, freq_remap, x.device)
    return _capi.rope_complex(x, cos, sin, n_rope, out_dtype=out_dtype)


class WanRMSNorm(nn.Module):
    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.dim = dim
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))

    def forward(self, x):
        return _capi.rmsnorm_rows(x, self.weight, self.eps)


class WanSelfAttention(nn.Module):
    """Same constructor / forward signature as the reference (:108-180)."""

    def __init__(self, dim, num_heads, window_size=(-1, -1), qk_norm=True, eps=1e-6, index=0, num_layers=0,
                 dtype=None, device=None):
        assert dim % num_heads == 0
        super().__init__()
        self.dim, self.num_heads, self.head_dim = dim, num_heads, dim // num_heads
        self.window_size, self.qk_norm, self.eps = window_size, qk_norm, eps
        self.index, self.num_layers = index, num_layers
        fk = dict(dtype=dtype, device=device)
        self.q, self.k, self.v, self.o = (nn.Linear(dim, dim, **fk) for _ in range(4))
        self.norm_q = WanRMSNorm(dim, eps=eps).to(device) if qk_norm else nn.Identity()
        self.norm_k = WanRMSNorm(dim, eps=eps).to(device) if qk_norm else nn.Identity()

    @torch.no_grad()
    def forward(self, x, seq_lens, grid_sizes, freqs, sa_drop_rate=0.0, per_block_tokens=128, p_remain_rates=0.8,
                freq_remap=None, block_neighbor_list=None, project=True):
        """project=False (jenga_amd.wan_dit's blocks): return the attention output BEFORE the output projection `self.o`,
        which the block then runs with the gate and the fp32 residual in its epilogue."""
        b, s, n, d = *x.shape[:2], self.num_heads, self.head_dim
        num_blocks = math.ceil(s / per_block_tokens)
        dense = sa_drop_rate <= 0.25
        if dense:
            # dense: every block kept (flash_attention with k_lens = seq_lens masks keys >= seq_len, :153-159)
            top_k, ffb = num_blocks, 0
        else:
            top_k = math.ceil(int(num_blocks * (1 - sa_drop_rate)))
            ffb = math.ceil(num_blocks // 21)
        nbm = None if dense else block_neighbor_list
        p = 2.0 if dense else p_remain_rates
        if x.is_cuda and b == 1 and d == 128 and per_block_tokens == 128 and isinstance(self.norm_q, WanRMSNorm) \
                and x.dtype in (torch.bfloat16, torch.float16):
            # one pass per tensor: WanRMSNorm (fp32 weight) + float64 RoPE + the op's bf16 cast, written straight into
            # buffers padded to whole 128-token blocks (the op's zero padding, ...diffres.py:448-451) -- no fp32
            # intermediates, no pad copies; V's GEMM writes into its padded buffer as well
            S_pad = num_blocks * 128
            cos, sin, n_rope = rope_tables(grid_sizes, freqs, freq_remap, x.device)
            qkv = torch.empty((3, 1, S_pad, self.dim), dtype=torch.bfloat16, device=x.device)
            if S_pad > s:
                qkv[:, :, s:].zero_()
            _capi.wan_norm_rope(self.q(x), self.norm_q.weight, cos, sin, min(n_rope, s), self.eps, out=qkv[0, 0])
            _capi.wan_norm_rope(self.k(x), self.norm_k.weight, cos, sin, min(n_rope, s), self.eps, out=qkv[1, 0])
            if x.dtype == torch.bfloat16:
                torch.addmm(self.v.bias, x[0], self.v.weight.t(), out=qkv[2, 0, :s])
            else:
                qkv[2, 0, :s] = self.v(x)[0].to(torch.bfloat16)
            if dense:
                seqlens = torch.clamp(seq_lens.to(device=x.device, dtype=torch.int32).reshape(-1), max=s)
            else:
                seqlens = torch.full((1,), s, dtype=torch.int32, device=x.device)
            q4, k4, v4 = (qkv[i].view(1, S_pad, n, d) for i in range(3))
            out = _op._combined(q4, k4, v4, top_k, seqlens, 0, 0.0, p, nbm, False, first_frame_blocks=ffb,
                                context_size=s)
            return self.o(out.to(x.dtype)) if project else out.to(x.dtype)
        # general path (any batch / head_dim the kernels take): the separate kernels and the padding op
        q = self.norm_q(self.q(x)).view(b, s, n, d)
        k = self.norm_k(self.k(x)).view(b, s, n, d)
        v = self.v(x).view(b, s, n, d)
        # rope_apply returns fp32 in the reference and the attention op immediately casts to bf16
        # (attention_block_triton_diffres.py:456-463 / flash_attention's half()): fuse that cast into the kernel.
        qr = rope_apply(q, grid_sizes, freqs, freq_remap, out_dtype=torch.bfloat16)
        kr = rope_apply(k, grid_sizes, freqs, freq_remap, out_dtype=torch.bfloat16)
        out = _op.block_sparse_attention_wan(qr, kr, v.to(torch.bfloat16), top_k, text_blocks=0,
                                             block_neighbor_list=nbm, p_remain_rates=p, first_frame_blocks=ffb,
                                             kv_lens=seq_lens if dense else None)   # k_lens mask: dense branch only
        return self.o(out.to(x.dtype).flatten(2)) if project else out.to(x.dtype).flatten(2)
