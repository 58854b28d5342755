"""ORACLE (test infrastructure): the reference's PyTorch-eager CPU path for AttenCarve, restated in torch.

The reference has no CPU implementation of its Triton kernel; its only CPU-capable attention is
`attention(mode="torch")` = F.scaled_dot_product_attention with an optional boolean mask
(/root/reference/hyvideo/modules/attenion.py:102-109), and its block selection is plain torch
(/root/reference/hyvideo/modules/attention_block_triton_diffres.py:198-295).  SURVEY.md §8(d) therefore defines "the
reference's PyTorch-CPU eager path" as: that selection + masked SDPA with the one-hot block mask expanded to 128x128
tiles.  This file restates both with torch ops (same ops in the same order as the cited lines; the index bookkeeping
of :253-276 is one scatter_ here) for two uses:
  * bench.py `cpu_baseline`: timed on the host cores next to the GPU number;
  * tests/test_oracle_golden.py: its mask must equal the numpy oracle's (which is pinned to the reference goldens).
Image query rows only see kept blocks and kv columns < seqlen; text rows see everything (flash_attn_func semantics,
:371-380).  Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import this package.
"""
import torch
import torch.nn.functional as F


def build_block_mask(q_img, k, top_k, text_start_block, num_blocks, prob_threshold, text_blocks, neighbors=None,
                     block=128, first_frame_blocks=0):
    """q_img [B,H,Sq,D], k [B,H,Sk,D] (any float dtype; the reference runs this in the tensor dtype) ->
    bool [B,H,nq,num_blocks].  Follows :216-293 line by line."""
    B, H, Sq, D = q_img.shape
    qp = q_img.reshape(B, H, -1, block, D).mean(dim=-2)                                   # :216
    kp = k.reshape(B, H, -1, block, D).mean(dim=-2)                                       # :217
    scores = torch.bmm(qp.reshape(B * H, -1, D), kp.reshape(B * H, -1, D).transpose(1, 2)) * (D ** -0.5)   # :221-227
    scores = scores.reshape(B, H, qp.shape[2], kp.shape[2])                               # :230-232
    probs = torch.softmax(scores[..., :text_start_block], dim=-1)                         # :235-238
    sorted_probs, indices = torch.sort(probs, dim=-1, descending=True)                    # :241
    cum = torch.cumsum(sorted_probs, dim=-1)                                              # :242
    n = torch.clamp((cum <= prob_threshold).sum(dim=-1) + 1, min=top_k)                   # :245-250
    nq = qp.shape[2]
    mask = torch.zeros((B, H, nq, num_blocks), dtype=torch.bool)
    rank = torch.arange(indices.shape[-1]).view(1, 1, 1, -1)
    sel = rank < n.unsqueeze(-1)                                                          # :265
    mask[..., :text_start_block].scatter_(-1, indices, sel)                               # :268-276
    if neighbors is not None:                                                             # :280-289
        nbm = torch.as_tensor(neighbors).bool()[:nq, :text_start_block]
        mask[:, :, :nbm.shape[0], :text_start_block] |= nbm[None, None]
    if first_frame_blocks > 0:                                  # wan/modules/attention_block_triton_diffres.py:400-406
        mask[:, :, :first_frame_blocks, :first_frame_blocks] = True
    if text_blocks > 0:                                                                   # :292-293
        mask[..., text_start_block:min(text_start_block + text_blocks, num_blocks)] = True
    return mask


def masked_attention(q, k, v, block_mask, seqlen, n_img_blocks, q_chunk_blocks=32, block=128, budget_s=None, clock=None):
    """q,k,v [1,H,S,D]; block_mask bool [1,H,nq_img,nb] -> o [1,H,S,D] via F.scaled_dot_product_attention with the
    mask expanded per 128x128 tile, processed in chunks of query blocks (the full [S,S] mask of one head at the 720p
    shape would be 13 GB).  Image rows: kept blocks & columns < seqlen; text rows: everything.
    With budget_s / clock (a time.perf_counter-like callable) the loop stops once the budget is used and returns
    (o, fraction_of_rows_done) -- bench.py's bounded sample."""
    _, H, S, D = q.shape
    nb = S // block
    o = torch.zeros_like(q)
    col_ok = (torch.arange(S) < seqlen).view(1, 1, 1, S)
    t0 = clock() if clock else None
    done = 0
    m0 = 0
    while m0 < nb:
        m1 = min(nb, m0 + q_chunk_blocks)
        rows = slice(m0 * block, m1 * block)
        if m0 < n_img_blocks:
            m1 = min(m1, n_img_blocks)
            rows = slice(m0 * block, m1 * block)
            bm = block_mask[:, :, m0:m1]                                                  # [1,H,c,nb]
            am = bm.repeat_interleave(block, dim=2).repeat_interleave(block, dim=3) & col_ok
        else:
            am = None                                                                     # text rows: no mask at all
        o[:, :, rows] = F.scaled_dot_product_attention(q[:, :, rows], k, v, attn_mask=am)
        done = m1
        m0 = m1
        if budget_s is not None and clock() - t0 > budget_s:
            break
    if budget_s is not None:
        return o, done / nb
    return o
