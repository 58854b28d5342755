"""ORACLE (test infrastructure): numpy restatement of the reference AttenCarve op.

Arrays are float32 holding values representable in the case dtype ("bfloat16" / "float16"); every place the
reference rounds to that dtype is an explicit rnd(...) here.

Follows /root/reference/hyvideo/modules/attention_block_triton_diffres.py
  build_block_mask        <- _build_block_index_with_importance_optimized   :198-295
                             (+ wan/modules/attention_block_triton_diffres.py:400-406 first_frame_blocks)
  sparse_rows             <- _triton_block_sparse_attn_fwd_kernel_onehot    :38-136, launcher :139-196
  text_rows               <- flash_attn_func call                           :371-380 (third-party flash-attn 2.6.3:
                             fp32 scores, softmax in fp32, P rounded to the input dtype before P.V, no length mask)
  block_sparse_attention  <- block_sparse_attention_combined / alias        :298-424 (HY); I2V/Wan pad+slice variants
                             hyvideo_i2v/...:323-328,385 and wan/...:448-463,519-532
"""

import numpy as np

from .rounding import rounder


# ----------------------------------------------------------------------------- selection (a8)
def pooled_scores(q_img, k, dtype, block=128):
    """q_img [B,H,Sq,D], k [B,H,Sk,D] -> scores [B,H,nq,nk] in `dtype` (:216-232)."""
    rnd = rounder(dtype)
    B, H, Sq, D = q_img.shape
    qp = rnd(q_img.reshape(B, H, Sq // block, block, D).astype(np.float32).mean(axis=-2, dtype=np.float32))
    kp = rnd(k.reshape(B, H, k.shape[2] // block, block, D).astype(np.float32).mean(axis=-2, dtype=np.float32))
    s = rnd(np.einsum("bhqd,bhkd->bhqk", qp, kp, dtype=np.float32))          # bmm output in dtype
    return rnd(s * np.float32(D ** -0.5))                                       # `* head_dim**-0.5` in dtype


def scores_from_pooled(qp, kp, dtype):
    """Pooled block means qp [B,H,nq,D], kp [B,H,nk,D] (dtype values) -> scores in dtype: the bmm output rounded to
    dtype, then `* head_dim**-0.5` in dtype (:221-232).  Lets a test feed the HIP pooling kernel's own output, which
    separates pooling-order noise from selection bugs."""
    rnd = rounder(dtype)
    s = rnd(np.einsum("bhqd,bhkd->bhqk", qp.astype(np.float32), kp.astype(np.float32), dtype=np.float32))
    return rnd(s * np.float32(qp.shape[-1] ** -0.5))


def build_block_mask_from_pooled(qp, kp, top_k, text_start_block, num_blocks, p, text_blocks, neighbors, dtype,
                                 first_frame_blocks=0):
    """build_block_mask with the pooling step (:216-217) already done."""
    B, H, nq, _ = qp.shape
    scores = scores_from_pooled(qp, kp, dtype)
    probs = row_probs(scores[..., :text_start_block], dtype)
    order, n = blocks_needed(probs, top_k, p, dtype)
    mask = np.zeros((B, H, nq, num_blocks), bool)
    rank = np.arange(order.shape[-1])
    sel = rank[None, None, None, :] < n[..., None]
    bi, hi, qi, ri = np.nonzero(sel)
    mask[bi, hi, qi, order[bi, hi, qi, ri]] = True
    if neighbors is not None:
        nbm = np.asarray(neighbors, bool)[:nq, :text_start_block]
        mask[:, :, :nbm.shape[0], :nbm.shape[1]] |= nbm[None, None]
    if first_frame_blocks > 0:
        mask[:, :, :first_frame_blocks, :first_frame_blocks] = True
    if text_blocks > 0 and text_start_block is not None:
        mask[:, :, :, text_start_block:min(text_start_block + text_blocks, num_blocks)] = True
    return mask, n


def row_probs(scores_img, dtype):
    """softmax over the image columns, result rounded to dtype (:238)."""
    rnd = rounder(dtype)
    x = scores_img.astype(np.float32)
    e = np.exp(x - x.max(axis=-1, keepdims=True))
    return rnd(e / e.sum(axis=-1, keepdims=True, dtype=np.float32))


def blocks_needed(probs, top_k, p, dtype):
    """-> (order, n): descending order (ties: lower index first -- the reference's torch.sort is unstable, any tie
    order is legal there) and n = max(#(cumsum <= p) + 1, top_k) with cumsum accumulated sequentially in fp32 and
    each partial rounded to dtype, compared against p rounded to dtype (:241-250)."""
    rnd = rounder(dtype)
    order = np.argsort(-probs, axis=-1, kind="stable")
    sp = np.take_along_axis(probs, order, axis=-1)
    cum = rnd(np.cumsum(sp.astype(np.float32), axis=-1, dtype=np.float32))
    thr = rnd(np.array([p], np.float32))[0]
    n = (cum <= thr).sum(axis=-1) + 1
    return order, np.maximum(n, top_k)


def build_block_mask(q_img, k, top_k, text_start_block, num_blocks, p, text_blocks, neighbors, dtype,
                     first_frame_blocks=0, block=128):
    """-> bool [B,H,nq,num_blocks]."""
    B, H, Sq, D = q_img.shape
    nq = Sq // block
    scores = pooled_scores(q_img, k, dtype, block)
    probs = row_probs(scores[..., :text_start_block], dtype)
    order, n = blocks_needed(probs, top_k, p, dtype)
    mask = np.zeros((B, H, nq, num_blocks), bool)
    rank = np.arange(order.shape[-1])
    sel = rank[None, None, None, :] < n[..., None]
    bi, hi, qi, ri = np.nonzero(sel)
    mask[bi, hi, qi, order[bi, hi, qi, ri]] = True                             # :253-276
    if neighbors is not None:                                                   # :280-289
        nbm = np.asarray(neighbors, bool)[:nq, :text_start_block]
        mask[:, :, :nbm.shape[0], :nbm.shape[1]] |= nbm[None, None]
    if first_frame_blocks > 0:                                                  # wan :400-406
        mask[:, :, :first_frame_blocks, :first_frame_blocks] = True
    if text_blocks > 0 and text_start_block is not None:                        # :292-293
        mask[:, :, :, text_start_block:min(text_start_block + text_blocks, num_blocks)] = True
    return mask


# ----------------------------------------------------------------------------- sparse kernel (a9)
def sparse_rows(q_img, k, v, seqlen, mask, sm_scale, dtype, text_amp=0.0, text_block_start=0, block=128):
    """q_img [B,H,Sq,D]; k,v [B,H,Sk,D]; mask bool [B,H,nq,nk]; seqlen int per batch (list) -> o [B,H,Sq,D]."""
    rnd = rounder(dtype)
    B, H, Sq, D = q_img.shape
    nq, nk = mask.shape[-2:]
    o = np.zeros_like(q_img, dtype=np.float32)                                  # torch.zeros_like(q) :156
    qk_scale = np.float32(sm_scale * 1.44269504)                                # :172, passed as an fp32 scalar
    amp = np.float32(text_amp)
    for b in range(B):
        sl = int(seqlen[b])
        for h in range(H):
            for m in range(nq):
                if m * block >= sl:                                             # :61-62
                    continue
                rows = np.arange(m * block, (m + 1) * block)
                qt = rnd(q_img[b, h, rows].astype(np.float32) * qk_scale)       # :87-88
                m_i = np.full(block, -np.inf, np.float32)
                l_i = np.zeros(block, np.float32)
                acc = np.zeros((block, D), np.float32)
                row_ok = rows < sl
                for j in range(nk):
                    if not mask[b, h, m, j]:
                        continue
                    cols = np.arange(j * block, (j + 1) * block)
                    s = qt @ k[b, h, cols].astype(np.float32).T                 # tl.dot, fp32 acc :110
                    s = np.where(row_ok[:, None], s, -np.inf)                   # :109
                    if j >= text_block_start:
                        s = s + amp                                             # :113-114
                    s = np.where((cols < sl)[None, :], s, -np.inf)              # :117-118
                    m_new = np.maximum(m_i, s.max(axis=1))
                    with np.errstate(invalid="ignore"):
                        alpha = np.exp2(m_i - m_new)
                        pmat = np.exp2(s - m_new[:, None])
                    acc = acc * alpha[:, None] + rnd(pmat) @ v[b, h, cols].astype(np.float32)   # :126-128
                    l_i = l_i * alpha + pmat.sum(axis=1, dtype=np.float32)
                    m_i = m_new
                with np.errstate(invalid="ignore", divide="ignore"):
                    res = rnd(acc / l_i[:, None])
                o[b, h, rows[row_ok]] = res[row_ok]                             # masked store :136
    return o


def text_rows(q_txt, k, v, sm_scale, dtype):
    """Dense attention of the text queries over ALL keys, no length mask (:371-380)."""
    rnd = rounder(dtype)
    s = np.einsum("bhqd,bhkd->bhqk", q_txt.astype(np.float32), k.astype(np.float32), dtype=np.float32)
    s = s * np.float32(sm_scale)
    pm = np.exp(s - s.max(axis=-1, keepdims=True))
    l = pm.sum(axis=-1, keepdims=True, dtype=np.float32)
    acc = np.einsum("bhqk,bhkd->bhqd", rnd(pm), v.astype(np.float32), dtype=np.float32)
    return rnd(acc / l)


def dense_varlen(q, k, v, cu_seqlens, sm_scale, dtype):
    """sa_drop_rate == 0 path: flash_attn_varlen_func over the segments in cu_seqlens on the flattened
    [B*S] token axis (attenion.py:108-121): tokens only attend inside their own segment."""
    rnd = rounder(dtype)
    B, H, S, D = q.shape
    qf = q.transpose(1, 0, 2, 3).reshape(H, B * S, D)
    kf = k.transpose(1, 0, 2, 3).reshape(H, B * S, D)
    vf = v.transpose(1, 0, 2, 3).reshape(H, B * S, D)
    of = np.zeros_like(qf, dtype=np.float32)
    for a, b_ in zip(cu_seqlens[:-1], cu_seqlens[1:]):
        a, b_ = int(a), int(b_)
        if b_ <= a:
            continue
        of[:, a:b_] = text_rows(qf[None, :, a:b_], kf[None, :, a:b_], vf[None, :, a:b_], sm_scale, dtype)[0]
    return of.reshape(H, B, S, D).transpose(1, 0, 2, 3)


# ----------------------------------------------------------------------------- whole op (a11)
def block_sparse_attention(query, key, value, top_k, dtype, cu_seqlens_q=None, text_blocks=2, text_amp=0.0,
                           block_neighbor_list=None, shape_xfuse=False, p_remain_rates=0.5, flavour="hy",
                           first_frame_blocks=0, block=128, return_mask=False, pooled=None):
    """query/key/value [B,S,H,D] float32 arrays holding `dtype` values.  flavour: "hy" | "i2v" | "wan".
    pooled = (qp [B,H,nimg,D], kp [B,H,nb,D]): block means to select from instead of pooling q / k here (:216-217) -- a
    test hands in the HIP pooling kernel's own output, so that a 1-ulp difference of a pooled mean (fp32 summation order)
    cannot flip a block of the selection and the comparison of the attention output can demand EVERY row."""
    q = np.transpose(query, (0, 2, 1, 3))
    k = np.transpose(key, (0, 2, 1, 3))
    v = np.transpose(value, (0, 2, 1, 3))
    B, H, S, D = q.shape
    pad = 0
    if flavour == "hy":
        if cu_seqlens_q is None:
            raise ValueError("HY flavour needs cu_seqlens (its no-cu_seqlens branch pads into unused variables)")
        if S % block:
            raise ValueError("HY flavour requires S % 128 == 0")
        seqlens = [int(cu_seqlens_q[1])] * 1 if B == 1 else None
        if seqlens is None:
            raise ValueError("reference uses cu_seqlens_q[1:2] -> batch 1 only")
    elif flavour == "i2v":
        seqlens = [int(cu_seqlens_q[1])]
        pad = (block - S % block) % block
    elif flavour == "wan":
        seqlens = [S] * B
        pad = (block - S % block) % block
    else:
        raise ValueError(flavour)
    if pad:
        z = np.zeros((B, H, pad, D), np.float32)
        q, k, v = (np.concatenate([t, z], axis=2) for t in (q, k, v))
    Sp = q.shape[2]
    nb = Sp // block
    nimg = nb - text_blocks
    sm_scale = D ** -0.5
    outs = []
    mask = None
    if nimg > 0:
        q_img = q[:, :, :nimg * block]
        if pooled is not None:
            mask, _ = build_block_mask_from_pooled(np.asarray(pooled[0], np.float32), np.asarray(pooled[1], np.float32),
                                                   top_k, nimg, nb, p_remain_rates, text_blocks, block_neighbor_list,
                                                   dtype, first_frame_blocks=first_frame_blocks)
        else:
            mask = build_block_mask(q_img, k, top_k, nimg, nb, p_remain_rates, text_blocks, block_neighbor_list, dtype,
                                    first_frame_blocks=first_frame_blocks, block=block)
        outs.append(sparse_rows(q_img, k, v, seqlens, mask, sm_scale, dtype, text_amp, nimg, block))
    if text_blocks > 0:
        outs.append(text_rows(q[:, :, nimg * block:], k, v, sm_scale, dtype))
    o = np.concatenate(outs, axis=2)[:, :, :S]
    o = np.transpose(o, (0, 2, 1, 3))
    if not shape_xfuse:
        o = o.reshape(B, S, H * D)
    return (o, mask) if return_mask else o
