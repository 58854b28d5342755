"""AttenCarve operator: drop-in for the reference's `block_sparse_attention`.

Mirrors (same names, argument order, defaults, return layout):
  hyvideo/modules/attention_block_triton_diffres.py:399-424      -> block_sparse_attention         (HunyuanVideo)
  hyvideo_i2v/modules/attention_block_triton_diffres.py:398-423  -> block_sparse_attention_i2v     (pads, text_blocks=4)
  wan/modules/attention_block_triton_diffres.py:535-562          -> block_sparse_attention_wan     (pads, bf16, first frame)

Everything numeric runs in libjenga_amd.so (HIP, gfx950) through jenga_amd._capi:
  jenga_block_pool + jenga_block_select  (reference :198-295)   selection -> ascending kept-block lists
  jenga_pack_v + jenga_bsattn_fwd        (reference :38-196, :371-380)   image rows + text rows in one launch
torch is used for allocation, views and zero-padding only.  No CPU fallback.
"""

import torch

from .. import _capi

BLOCK = 128


def _seqlens_from_cu(cu_seqlens_q, device):
    # reference :327-329: seqlens = cu_seqlens_q[1:2] (only batch element 0's valid length is ever used)
    s = cu_seqlens_q[1:2]
    return s.to(device=device, dtype=torch.int32)


def build_block_index(query, key, top_k, text_blocks, prob_threshold, block_neighbor_list=None,
                      first_frame_blocks=0, want_mask=False, pooled=None, head_dim=128):
    """query/key [B,S,H,128] (S multiple of 128).  -> (mask|None, idx, cnt) for the image query blocks.
    Replaces _build_block_index_with_importance_optimized (:198-295).  pooled = (qpool [B,H,nimg,128],
    kpool [B,H,nb,128]) when the caller's norm+RoPE kernel already produced the block means.  head_dim < 128: the
    tensors are zero-padded to 128 channels, the scores scaled by head_dim ** -0.5 (:232)."""
    B, S, H, D = query.shape
    nb = S // BLOCK
    nimg = nb - text_blocks
    if pooled is not None:
        qpool, kpool = pooled
    else:
        qpool = _capi.block_pool(query, nimg)
        kpool = _capi.block_pool(key, nb)
    return _capi.block_select(qpool, kpool, block_neighbor_list, nimg, text_blocks, top_k, prob_threshold,
                              first_frame_blocks=first_frame_blocks, want_mask=want_mask, head_dim=head_dim)


def attencarve_packed(q, k, vt, top_k, seqlens, text_blocks, text_amp, prob_threshold, block_neighbor_list,
                      first_frame_blocks=0, out=None, return_lists=False, pooled=None):
    """Core used by the DiT blocks: q/k [B,S,H,128] already normed+roped (any strides), vt = packed V workspace
    (jenga_pack_v), seqlens int32 [B] on the device.  Selection + one attention launch; returns o [B,S,H,128]
    (written into `out` if given, which may be a strided view, e.g. the left part of the single-stream blocks'
    concat buffer)."""
    B, S, H, D = q.shape
    nb = S // BLOCK
    nimg = nb - text_blocks
    idx = cnt = None
    if nimg > 0:
        _, idx, cnt = build_block_index(q, k, top_k, text_blocks, prob_threshold, block_neighbor_list,
                                        first_frame_blocks, pooled=pooled)
    o = _capi.bsattn_fwd(q, k, vt, seqlens, idx, cnt, nimg, D ** -0.5, text_amp, nimg, out=out)
    return (o, idx, cnt) if return_lists else o


def _combined(query, key, value, top_k, seqlens, text_blocks, text_amp, prob_threshold, block_neighbor_list,
              shape_xfuse, first_frame_blocks=0, context_size=None, return_mask=False):
    B, S, H, D = query.shape
    if D not in (16, 32, 64, 128):         # the Triton kernel's own assert (:155)
        raise ValueError(f"jenga_amd: head_dim must be 16, 32, 64 or 128 (got {D})")
    head_dim = D
    if D != 128:
        # Narrow heads run on the 128-channel kernels with zero channels appended: the extra products are exact zeros in
        # every dot product (pooled scores, QK^T), the extra output channels are dropped, and the two scales keep the true
        # head_dim ** -0.5.  No entry script of the reference reaches this (all three models have 128-channel heads), so
        # it is kept simple rather than fast: 128 / D times the FLOPs.
        query, key, value = (torch.nn.functional.pad(t, [0, 128 - D]) for t in (query, key, value))
        D = 128
    if S % BLOCK:
        raise ValueError(f"jenga_amd: sequence length {S} is not a multiple of {BLOCK}")
    nb = S // BLOCK
    nimg = nb - text_blocks
    if nimg < 0:
        raise ValueError("text_blocks exceeds the number of blocks")
    mask = idx = cnt = None
    if nimg > 0:
        mask, idx, cnt = build_block_index(query, key, top_k, text_blocks, prob_threshold, block_neighbor_list,
                                           first_frame_blocks, want_mask=return_mask, head_dim=head_dim)
    vt = _capi.pack_v(value, nb)
    out = _capi.bsattn_fwd(query, key, vt, seqlens, idx, cnt, nimg, head_dim ** -0.5, text_amp, nimg)
    if context_size is not None and context_size != S:
        out = out[:, :context_size]
    if head_dim != 128:
        out = out[..., :head_dim].contiguous()
        D = head_dim
    if not shape_xfuse:
        out = out.reshape(B, out.shape[1], H * D)
    return (out, mask) if return_mask else out


def _check_inputs(query, key, value):
    if query.dtype not in (torch.bfloat16, torch.float16):
        raise ValueError(f"jenga_amd: dtype must be bfloat16 or float16, got {query.dtype}")
    if key.dtype != query.dtype or value.dtype != query.dtype:
        raise ValueError("jenga_amd: query/key/value dtypes differ")
    if query.dim() != 4 or key.shape != value.shape or query.shape != key.shape:
        raise ValueError("jenga_amd: expected query/key/value of identical shape [B,S,H,D]")


def _unit_inner(t):
    return t if t.stride(-1) == 1 and all(s % 8 == 0 for s in t.stride()[:-1]) else t.contiguous()


def block_sparse_attention(
    query: torch.Tensor,
    key: torch.Tensor,
    value: torch.Tensor,
    top_k: int,
    block_size_M: int = 128,
    block_size_N: int = 128,
    cu_seqlens_q: torch.Tensor = None,
    cu_seqlens_kv: torch.Tensor = None,
    max_seqlen_q: int = None,
    max_seqlen_kv: int = None,
    text_blocks: int = 2,
    text_amp: float = 0.0,
    block_neighbor_list: torch.Tensor = None,
    shape_xfuse: bool = False,
    p_remain_rates: float = 0.5,
    return_mask: bool = False,
):
    """HunyuanVideo flavour.  q/k/v [B,S,H,128], S % 128 == 0 (the reference's padding branch is dead code, :331-336),
    cu_seqlens_q from get_cu_seqlens (only element [1] is used).  Returns [B,S,H*D], or [B,S,H,D] if shape_xfuse."""
    _check_inputs(query, key, value)
    if block_size_M != BLOCK or block_size_N != BLOCK:
        raise ValueError("jenga_amd: block sizes must be 128")
    if cu_seqlens_q is None or cu_seqlens_kv is None:
        raise ValueError("jenga_amd (HunyuanVideo flavour): cu_seqlens_q / cu_seqlens_kv are required")
    if query.shape[0] != 1:
        raise ValueError("jenga_amd (HunyuanVideo flavour): batch must be 1 (the reference reads cu_seqlens_q[1:2] only)")
    query, key, value = _unit_inner(query), _unit_inner(key), _unit_inner(value)
    seqlens = _seqlens_from_cu(cu_seqlens_q, query.device)
    return _combined(query, key, value, top_k, seqlens, text_blocks, text_amp, p_remain_rates, block_neighbor_list,
                     shape_xfuse, return_mask=return_mask)


def _pad_seq(t, pad):
    return torch.nn.functional.pad(t, [0, 0, 0, 0, 0, pad]) if pad else t


def block_sparse_attention_i2v(query, key, value, top_k, block_size_M=128, block_size_N=128, cu_seqlens_q=None,
                               cu_seqlens_kv=None, max_seqlen_q=None, max_seqlen_kv=None, text_blocks=4,
                               text_amp=0.0, block_neighbor_list=None, shape_xfuse=False, p_remain_rates=0.5,
                               return_mask=False):
    """HunyuanVideo-I2V flavour: pads S to a multiple of 128, text_blocks defaults to 4, output sliced back."""
    _check_inputs(query, key, value)
    if cu_seqlens_q is None:
        raise ValueError("jenga_amd (I2V flavour): cu_seqlens_q is required")
    B, S, H, D = query.shape
    pad = (BLOCK - S % BLOCK) % BLOCK
    q, k, v = (_pad_seq(_unit_inner(t), pad) for t in (query, key, value))
    seqlens = _seqlens_from_cu(cu_seqlens_q, query.device)
    return _combined(q, k, v, top_k, seqlens, text_blocks, text_amp, p_remain_rates, block_neighbor_list,
                     shape_xfuse, context_size=S, return_mask=return_mask)


def block_sparse_attention_wan(query, key, value, top_k, block_size_M=128, block_size_N=128, cu_seqlens_q=None,
                               cu_seqlens_kv=None, max_seqlen_q=None, max_seqlen_kv=None, text_blocks=0,
                               text_amp=0.0, block_neighbor_list=None, shape_xfuse=False, p_remain_rates=0.9,
                               first_frame_blocks=0, return_mask=False, kv_lens=None):
    """Wan flavour: always pads, computes in bf16 whatever the input dtype, seqlen = unpadded S, dense first-frame
    rule, returns the input dtype.  kv_lens (int32 [B] tensor, jenga_amd extension): keys at or beyond it are masked
    -- what WanSelfAttention's dense branch gets from flash_attention(k_lens=seq_lens) (wan/modules/model_mul.py:153-159)
    when teacache_forward padded the tokens beyond the real sequence; the reference's sparse branch has no such mask."""
    out_dtype = query.dtype
    B, S, H, D = query.shape
    pad = (BLOCK - S % BLOCK) % BLOCK
    q, k, v = (_pad_seq(_unit_inner(t.to(torch.bfloat16)), pad) for t in (query, key, value))
    if kv_lens is not None:
        seqlens = torch.clamp(torch.as_tensor(kv_lens).to(device=query.device, dtype=torch.int32).reshape(-1), max=S)
        if seqlens.numel() != B:
            raise ValueError("kv_lens must have one entry per batch element")
    else:
        seqlens = torch.full((B,), S, dtype=torch.int32, device=query.device)
    res = _combined(q, k, v, top_k, seqlens, text_blocks, text_amp, p_remain_rates, block_neighbor_list,
                    shape_xfuse, first_frame_blocks=first_frame_blocks, context_size=S, return_mask=return_mask)
    if return_mask:
        return res[0].to(out_dtype), res[1]
    return res.to(out_dtype)
