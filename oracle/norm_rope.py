"""ORACLE (test infrastructure): numpy restatement of per-head RMSNorm and 3-axis RoPE.

Follows /root/reference/hyvideo/modules/norm_layers.py:43,56-59 (RMSNorm) and
/root/reference/hyvideo/modules/posemb_layers.py:14-53 (meshgrid), :133-137 (rotate_half), :181-229
(apply_rotary_emb, real cos/sin branch), :232-351 (table construction)."""
import numpy as np

from .rounding import rounder


def rmsnorm(x, weight, dtype, eps=1e-6):
    """x [..., D] float32 holding dtype values -> (x.float()*rsqrt(mean(x^2)+eps)).type_as(x) * weight, in dtype."""
    rnd = rounder(dtype)
    xf = x.astype(np.float32)
    ms = (xf * xf).mean(axis=-1, keepdims=True, dtype=np.float32)
    y = rnd(xf * (np.float32(1.0) / np.sqrt(ms + np.float32(eps), dtype=np.float32)))
    if weight is None:
        return y
    return rnd(y * weight.astype(np.float32))


def rope_tables(rope_dim_list, sizes, theta=256.0):
    """get_nd_rotary_pos_embed(rope_dim_list, sizes, theta, use_real=True) -> cos, sin [prod(sizes), sum(dims)] fp32."""
    axes = [np.arange(n, dtype=np.float32) for n in sizes]     # linspace(0, n, n+1)[:n] == arange for these sizes
    grid = np.meshgrid(*axes, indexing="ij")
    cos_parts, sin_parts = [], []
    for dim, g in zip(rope_dim_list, grid):
        expo = np.arange(0, dim, 2, dtype=np.float32)[: dim // 2] / np.float32(dim)
        # torch computes theta**expo with its own fp32 powf; libm differences are ulps, so go through fp64 and round
        freqs = (np.float32(1.0) / np.power(np.float64(theta), expo.astype(np.float64)).astype(np.float32)).astype(np.float32)
        ang = np.outer(g.reshape(-1).astype(np.float32), freqs).astype(np.float32)
        cos_parts.append(np.repeat(np.cos(ang, dtype=np.float32), 2, axis=1))
        sin_parts.append(np.repeat(np.sin(ang, dtype=np.float32), 2, axis=1))
    return np.concatenate(cos_parts, axis=1), np.concatenate(sin_parts, axis=1)


def apply_rotary_emb(x, cos, sin, dtype):
    """x [B,S,H,D]; cos,sin [S,D] fp32: (x*cos + rotate_half(x)*sin) in fp32, separate roundings, -> dtype."""
    rnd = rounder(dtype)
    xf = x.astype(np.float32)
    rot = np.empty_like(xf)
    rot[..., 0::2] = -xf[..., 1::2]
    rot[..., 1::2] = xf[..., 0::2]
    c = cos[None, :, None, :].astype(np.float32)
    s = sin[None, :, None, :].astype(np.float32)
    return rnd((xf * c).astype(np.float32) + (rot * s).astype(np.float32))
