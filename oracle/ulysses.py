"""ORACLE (test infrastructure): the Ulysses sequence-parallel contract, restated.  PARITY UNPINNED: the reference
wrapper (hyvideo/modules/xdit_ring_atten.py:61-222) subclasses yunchang==0.6.3.post1's LongContextAttention and calls
yunchang.comm.all_to_all.SeqAllToAll4D (neither is in /root/reference nor installed; the reference has no tests and no
CPU path for it), so this file restates the published Ulysses algorithm anchored on the reference's call sites:

  * scatter_idx=2 / gather_idx=1 (:26-27,118-131): heads are split into N contiguous slices [r*H/N, (r+1)*H/N)
    (consistent with the explicit slicing of the text K/V at :159-175); sequences are concatenated rank-major;
  * text Q after the all-to-all is truncated to the first S_txt rows (:129-131) = rank 0's text rows for this rank's
    heads = this rank's own (replicated) text rows;
  * cu_seqlens = [0, n_valid_text + N*S_loc, N*S_loc + S_txt] (:105,183-184); top_k is passed through (the caller has
    already multiplied it by N, models_mul_block_gc_ha_multigpu.py:249-251);
  * the op runs with shape_xfuse=True on [1, N*S_loc + S_txt, H/N, D] (:186-199);
  * image output goes back with the inverse all-to-all, text output is the concatenation over head slices (:206-217).

simulate(...) runs all N ranks in one process with a fake all-to-all and returns each rank's [1, S_loc+S_txt, H, D].
"""
import numpy as np

from . import attention as oa


def simulate(q_shards, k_shards, v_shards, q_txt, k_txt, v_txt, top_k, n_valid_text, dtype, text_amp=0.0,
             neighbors=None, p=0.3):
    """q_shards[r]: [1,S_loc,H,D] (rank r's slice of the Hilbert-ordered image tokens); *_txt [1,S_txt,H,D]."""
    N = len(q_shards)
    _, S_loc, H, D = q_shards[0].shape
    Hn = H // N
    S_txt = q_txt.shape[1]
    S_img = N * S_loc
    outs = []
    for r in range(N):
        hs = slice(r * Hn, (r + 1) * Hn)
        cat = lambda shards, txt: np.concatenate([s[:, :, hs] for s in shards] + [txt[:, :, hs]], axis=1)
        q, k, v = cat(q_shards, q_txt), cat(k_shards, k_txt), cat(v_shards, v_txt)
        cu = np.array([0, n_valid_text + S_img, S_img + S_txt], np.int64)
        outs.append(oa.block_sparse_attention(q, k, v, top_k, dtype, cu_seqlens_q=cu, text_blocks=S_txt // 128,
                                              text_amp=text_amp, block_neighbor_list=neighbors, shape_xfuse=True,
                                              p_remain_rates=p))          # [1, S, Hn, D]
    results = []
    for r in range(N):
        img = np.concatenate([outs[p_][:, r * S_loc:(r + 1) * S_loc] for p_ in range(N)], axis=2)   # heads back
        txt = np.concatenate([outs[p_][:, S_img:] for p_ in range(N)], axis=2)
        results.append(np.concatenate([img, txt], axis=1))
    return results


def pack_heads(t, n_ranks):
    """Stand-in for jenga_ulysses_pack_heads on CPU tensors (tests only): [B,S_loc,H,D] -> peer-major
    [N,B,S_loc,H/N,D], rank r receives the contiguous head slice [r*H/N, (r+1)*H/N)."""
    B, S, H, D = t.shape
    return t.reshape(B, S, n_ranks, H // n_ranks, D).permute(2, 0, 1, 3, 4).contiguous()


def unpack_heads(recv, n_ranks, out):
    """Stand-in for jenga_ulysses_unpack_heads on CPU tensors (tests only): peer-major [N,B,S_loc,H/N,D] -> out
    [B,S_loc,H,D]."""
    Np, B, S, Hn, D = recv.shape
    out.copy_(recv.permute(1, 2, 0, 3, 4).reshape(B, S, Np * Hn, D))
    return out



def qkv_prologue(xq, xk, xv, wq, wk, cos, sin, outs, heads_per_peer, head0, n_heads, s_rope, dtype="bfloat16"):
    """Stand-in for jenga_sp_qkv_prologue on CPU tensors (tests only): per-head RMSNorm of q, k
    (norm_layers.py:56-59), RoPE on tokens < s_rope (posemb_layers.py:181-229), then head h of q, k, v goes to
    outs[i][h // heads_per_peer, ..., h % heads_per_peer, :] (peer-major outputs [N,B,S,Hn,D]) or to
    outs[i][..., h - head0, :] (head-window outputs [B,S,n_heads,D]) -- xdit_ring_atten.py:118-131 / :159-175."""
    import torch

    from . import norm_rope as onr
    tn = lambda t: t.detach().float().numpy()
    res = []
    # (xq, xk) or xv may be None together with their outputs: the blocks post Q, K before the V GEMM has run
    for x, w in ((xq, wq), (xk, wk)):
        if x is None:
            res.append(None)
            continue
        y = onr.rmsnorm(tn(x), None if w is None else tn(w), dtype)
        if cos is not None and s_rope > 0:
            y[:, :s_rope] = onr.apply_rotary_emb(y[:, :s_rope], tn(cos)[:s_rope], tn(sin)[:s_rope], dtype)
        res.append(torch.from_numpy(y).to(x.dtype))
    res.append(xv)
    for o, y in zip(outs, res):
        if y is None:
            continue
        y = y[:, :, head0:head0 + n_heads]
        if o.dim() == 5:
            o.copy_(pack_heads(y, o.shape[0]))
        else:
            o.copy_(y)
