"""RoPE for the HunyuanVideo DiT, drop-in for hyvideo/modules/posemb_layers.py.

get_meshgrid_nd / get_1d_rotary_pos_embed / get_nd_rotary_pos_embed build the (cos, sin) tables with the same torch
calls as the reference (:14-53, :232-351) -- they run once per resolution stage on the host and are static inputs;
`apply_rotary_emb` (:181-229, real cos/sin branch) and the fused `qk_norm_rope` run the HIP kernel
jenga_rmsnorm_rope.
"""
from typing import List, Tuple, Union

import torch

from .. import _capi


def _to_tuple(x, dim=2):
    if isinstance(x, int):
        return (x,) * dim
    if len(x) == dim:
        return tuple(x)
    raise ValueError(f"Expected length {dim} or int, but got {x}")


def get_meshgrid_nd(start, *args, dim=2):
    """n-D grid of positions, [dim, *num]; argument convention of the reference (:14-53)."""
    if len(args) == 0:
        num, start, stop = _to_tuple(start, dim), (0,) * dim, _to_tuple(start, dim)
    elif len(args) == 1:
        start, stop = _to_tuple(start, dim), _to_tuple(args[0], dim)
        num = [stop[i] - start[i] for i in range(dim)]
    elif len(args) == 2:
        start, stop, num = _to_tuple(start, dim), _to_tuple(args[0], dim), _to_tuple(args[1], dim)
    else:
        raise ValueError(f"len(args) should be 0, 1 or 2, but got {len(args)}")
    axes = [torch.linspace(start[i], stop[i], num[i] + 1, dtype=torch.float32)[: num[i]] for i in range(dim)]
    return torch.stack(torch.meshgrid(*axes, indexing="ij"), dim=0)


def get_1d_rotary_pos_embed(dim: int, pos, theta: float = 10000.0, use_real: bool = False,
                            theta_rescale_factor: float = 1.0, interpolation_factor: float = 1.0):
    if isinstance(pos, int):
        pos = torch.arange(pos).float()
    if theta_rescale_factor != 1.0:
        theta *= theta_rescale_factor ** (dim / (dim - 2))
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2)[: (dim // 2)].float() / dim))
    freqs = torch.outer(pos * interpolation_factor, freqs)
    if use_real:
        return freqs.cos().repeat_interleave(2, dim=1), freqs.sin().repeat_interleave(2, dim=1)
    return torch.polar(torch.ones_like(freqs), freqs)


def get_nd_rotary_pos_embed(rope_dim_list, start, *args, theta=10000.0, use_real=False,
                            theta_rescale_factor: Union[float, List[float]] = 1.0,
                            interpolation_factor: Union[float, List[float]] = 1.0):
    grid = get_meshgrid_nd(start, *args, dim=len(rope_dim_list))
    n = len(rope_dim_list)

    def _listify(v, name):
        if isinstance(v, (int, float)):
            return [v] * n
        if isinstance(v, list) and len(v) == 1:
            return [v[0]] * n
        assert len(v) == n, f"len({name}) should equal to len(rope_dim_list)"
        return v

    trf, itf = _listify(theta_rescale_factor, "theta_rescale_factor"), _listify(interpolation_factor, "interpolation_factor")
    embs = [get_1d_rotary_pos_embed(rope_dim_list[i], grid[i].reshape(-1), theta, use_real=use_real,
                                    theta_rescale_factor=trf[i], interpolation_factor=itf[i]) for i in range(n)]
    if use_real:
        return torch.cat([e[0] for e in embs], dim=1), torch.cat([e[1] for e in embs], dim=1)
    return torch.cat(embs, dim=1)


def _tables(freqs_cis, x, head_first=False):
    """-> (cos, sin) fp32 [S, D] on x's device.  freqs_cis: the real (cos, sin) pair HunyuanVideo uses, or the COMPLEX table of
    posemb_layers.py:216-227 ([S, D/2]): (a + ib)(c + is) = (ac - bs) + i(as + bc) is the real form with cos = Re, sin = Im
    repeated per pair -- the same two products and one add in fp32 (tests/test_oracle_golden.py pins that bit for bit)."""
    S = x.shape[-2] if head_first else x.shape[1]
    if isinstance(freqs_cis, tuple):
        cos, sin = freqs_cis
    else:
        if not torch.is_complex(freqs_cis):
            raise ValueError("jenga_amd: freqs_cis must be a (cos, sin) tuple or a complex tensor")
        cos = freqs_cis.real.repeat_interleave(2, dim=-1)
        sin = freqs_cis.imag.repeat_interleave(2, dim=-1)
    assert cos.shape == (S, x.shape[-1]), f"freqs_cis shape {tuple(cos.shape)} does not match x shape {tuple(x.shape)}"
    return cos.to(x.device, torch.float32).contiguous(), sin.to(x.device, torch.float32).contiguous()


def _rope(x, cos, sin, head_first):
    if not head_first:
        return _capi.rmsnorm_rope(x, None, cos, sin, eps=-1.0)
    # [B, H, S, D]: the kernel takes any (batch, token, head) strides -- run it on the transposed view, write a [B, H, S, D] result
    out = torch.empty_like(x, memory_format=torch.contiguous_format)
    _capi.rmsnorm_rope(x.transpose(1, 2), None, cos, sin, eps=-1.0, out=out.transpose(1, 2))
    return out


def apply_rotary_emb(xq: torch.Tensor, xk: torch.Tensor, freqs_cis, head_first: bool = False
                     ) -> Tuple[torch.Tensor, torch.Tensor]:
    """xq, xk [B,S,H,128] ([B,H,S,128] with head_first); freqs_cis = (cos, sin) fp32 [S,128] or complex [S,64]:
    (x*cos + rotate_half(x)*sin).type_as(x)  (posemb_layers.py:181-229, both branches)."""
    cos, sin = _tables(freqs_cis, xq, head_first)
    return _rope(xq, cos, sin, head_first), _rope(xk, cos, sin, head_first)


def apply_rotary_emb_single(xq, freqs_cis, head_first=False):
    cos, sin = _tables(freqs_cis, xq, head_first)
    return _rope(xq, cos, sin, head_first)


def qk_norm_rope(q, k, q_weight, k_weight, freqs_cis=None, eps=1e-6, out_q=None, out_k=None):
    """Fused per-head RMSNorm (+RoPE when freqs_cis is given) for a Q/K pair, one pass each over HBM.
    Equivalent to `apply_rotary_emb(q_norm(q), k_norm(k), freqs_cis)` of the reference blocks
    (models_mul_block_gc_ha_multigpu.py:205-214)."""
    cos = sin = None
    if freqs_cis is not None:
        cos, sin = _tables(freqs_cis, q)
    return (_capi.rmsnorm_rope(q, q_weight, cos, sin, eps=eps, out=out_q),
            _capi.rmsnorm_rope(k, k_weight, cos, sin, eps=eps, out=out_k))
