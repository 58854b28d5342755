"""ORACLE helper: explicit low-precision rounding points on float32 numpy arrays."""
import numpy as np


def round_bf16(x):
    """fp32 -> nearest-even bf16, returned as fp32 (same as torch .to(bfloat16).float())."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    nan = np.isnan(x)
    r = ((u.astype(np.uint64) + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    out = r.view(np.float32).copy()
    out[nan] = np.nan
    return out.reshape(x.shape)


def round_fp16(x):
    return np.asarray(x, dtype=np.float32).astype(np.float16).astype(np.float32)


def rounder(dtype):
    if dtype in ("bfloat16", "bf16"):
        return round_bf16
    if dtype in ("float16", "fp16"):
        return round_fp16
    raise ValueError(dtype)


def bf16_bits_to_f32(u16):
    return (np.asarray(u16, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


def f32_to_bf16_bits(x):
    return (round_bf16(x).view(np.uint32) >> 16).astype(np.uint16)
