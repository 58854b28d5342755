"""HunyuanVideo DiT with the Jenga forward, as the AttenCarve hot path's CALLER.

Counterpart of
  hyvideo/modules/models_mul_block_gc_ha_multigpu.py   MMDoubleStreamBlock :28-316, MMSingleStreamBlock :319-500,
                                                        HYVideoDiffusionTransformer :503-845 (config :852-870)
  jenga_hyvideo.py                                      ra_forward :61-234 (Hilbert gather/scatter + step-skip cache)
  jenga_hyvideo_multigpu.py                             new_forward / transformer_sub_forward :109-331 (sequence shards)

The blocks keep the reference's positional calling convention (12 arguments, jenga_hyvideo.py:143-156 / :162-175) and
state-dict names of the layers they own.  The dense linear algebra (QKV / proj / MLP GEMMs, LayerNorm, adaLN
modulation, GELU) is plain torch -> hipBLASLt / rocm-torch: out of the hot-path scope.  Everything
on the hot path is the HIP library: per-head RMSNorm+RoPE written straight into the concatenated (image|text) Q/K
buffers, V re-tiling straight from the GEMM outputs, block selection, block-sparse attention writing straight into
the consumer's buffer, Hilbert gather/scatter.

Weights are synthetic (random init; there are no checkpoints in this environment); the time/text embedders of the
real model (timestep MLPs, token refiner) are replaced by one linear each -- they run once per step on <= 256 tokens.
"""
import math
import os
from typing import Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _capi
from . import gilbert as G
from .modules import attention_block_sparse as op
from .modules.attention import attention as dense_attention
from .modules.attention import get_cu_seqlens, my_parallel_attention
from .modules.norm_layers import RMSNorm
from .modules.posemb_layers import get_nd_rotary_pos_embed
from .modules import ulysses

# jenga_hyvideo.py:28
NON_SKIP_STEPS = [0, 1, 2, 3, 4, 7, 10, 13, 16, 19, 22, 25, 26, 29, 32, 35, 38, 41, 43, 45, 46, 47, 49]

HUNYUAN_VIDEO_CONFIG = {  # models_mul_block_gc_ha_multigpu.py:852-870
    "HYVideo-T/2-cfgdistill": dict(mm_double_blocks_depth=20, mm_single_blocks_depth=40, rope_dim_list=[16, 56, 56],
                                   hidden_size=3072, heads_num=24, mlp_width_ratio=4, guidance_embed=True),
}


def modulate(x, shift=None, scale=None):
    if scale is None and shift is None:
        return x
    if shift is None:
        return x * (1 + scale.unsqueeze(1))
    if scale is None:
        return x + shift.unsqueeze(1)
    return torch.addcmul(shift.unsqueeze(1), x, 1 + scale.unsqueeze(1))


def apply_gate(x, gate=None, tanh=False):
    if gate is None:
        return x
    return x * (gate.unsqueeze(1).tanh() if tanh else gate.unsqueeze(1))


class ModulateDiT(nn.Module):
    def __init__(self, hidden_size, factor, dtype=None, device=None):
        super().__init__()
        self.act = nn.SiLU()
        self.linear = nn.Linear(hidden_size, factor * hidden_size, bias=True, dtype=dtype, device=device)

    def forward(self, x):
        return self.linear(self.act(x))


class MLP(nn.Module):
    def __init__(self, in_channels, hidden_channels, dtype=None, device=None):
        super().__init__()
        self.fc1 = nn.Linear(in_channels, hidden_channels, bias=True, dtype=dtype, device=device)
        self.fc2 = nn.Linear(hidden_channels, in_channels, bias=True, dtype=dtype, device=device)

    def hidden(self, x):
        # fc1 + bias + tanh-GELU in ONE hipBLASLt call (jenga_linear: GELU epilogue on the fp32 accumulator): the separate
        # activation pass was 5.7 GB of HBM traffic per layer at the 720p shape
        return _capi.linear(x, self.fc1.weight, self.fc1.bias, act=_capi.ACT_GELU_TANH)

    def forward(self, x):
        return self.fc2(self.hidden(x))


FUSE_GATE_RESIDUAL = os.environ.get("JENGA_FUSE_GATE", "1") != "0"
SPLIT_LINEAR1 = os.environ.get("JENGA_SPLIT_LINEAR1", "1") != "0"
GATED_BIAS_F32 = os.environ.get("JENGA_GATED_BIAS", "f32") != "bf16"
# Sequence parallel, round 4: exchange / compute overlap (the reference issues everything on one stream,
# xdit_ring_atten.py:118-131, 212-217).  Single-stream blocks: the Q, K, V exchange is posted right behind the QKV half of
# linear1 and the MLP half (GEMM + GELU, 57 % of the block's GEMM FLOPs) is issued while it is in flight; the last
# SP_MLP_TAIL share of the MLP columns is held back and issued behind the attention, under the O exchange.  Double-stream
# blocks: the Q|K GEMM is followed by the Q, K exchange, the V GEMM and the whole text stream run under it.
SP_OVERLAP = os.environ.get("JENGA_SP_OVERLAP", "1") != "0"
SP_MLP_TAIL = float(os.environ.get("JENGA_SP_MLP_TAIL", "0.375"))


# Round 6 (review item 2): the HBM-bound row kernels between the QKV GEMM and the attention launch -- fused RMSNorm + RoPE +
# pooling, V re-tiling, block selection (~1.15 ms per layer at the 720p shape, all serial behind MFMA-bound kernels until now) --
# run on a second stream beside the block's independent GEMM: in the single-stream blocks the MLP half of linear1 (+ GELU),
# which needs only the modulated input (models_mul_block_gc_ha_multigpu.py:392-500: `linear1` feeds qkv AND mlp); in the
# double-stream blocks the text stream's modulate + QKV GEMM + norm (:161-316).  Same kernels on the same inputs: results are
# bit-identical to the one-stream order (tests/test_gpu_dit.py).  Concurrent streams are safe since round 5's packed-fp32 fix
# (DESIGN.md section 4).  MEASURED: it loses -- 78.68 / 78.71 against 78.50 / 78.56 s/video on one box, two interleaved passes
# (profiles/r06_rowops_overlap_ab.json): on the power-capped board the GEMM beside the row kernels slows down by what they
# hide.  So the one-stream order stays the default; JENGA_ROWOPS_OVERLAP=1 is the opt-in.
ROWOPS_OVERLAP = os.environ.get("JENGA_ROWOPS_OVERLAP", "0") == "1"
_SIDE_STREAMS = {}


class _Fork:
    """fn() on the device's side stream, ordered behind everything enqueued on the current stream so far; .join() orders the
    current stream behind it and returns fn's result.  Tensors fn allocates belong to the side stream's pool and are consumed
    on the main stream after join(): the next fork starts behind a LATER point of the main stream, so a reused block is never
    written while the main stream still reads it; tensors of the main stream that fn reads are released by the caller after
    join()."""

    def __init__(self, device, fn):
        self.main = torch.cuda.current_stream(device)
        key = (device.index, self.main.cuda_stream)
        if key not in _SIDE_STREAMS:
            _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
        self.side = _SIDE_STREAMS[key]
        self.side.wait_stream(self.main)
        with torch.cuda.stream(self.side):
            self.result = fn()

    def join(self):
        self.main.wait_stream(self.side)
        return self.result


def linear_gate_residual(lin, x, gate, res, gate2=None, mask=None):
    """res + apply_gate(lin(x), gate) (models_mul...:297-315, 500).  Without a token mask the gate multiply and the
    residual add ride in the GEMM's epilogue (jenga_linear: per-channel gate = alpha vector, residual = C matrix; the
    bias is pre-multiplied by the gate): one pass over the output instead of three.  With the I2V token_replace mask
    (rows choose between two gates) the separate kernel stays.
    Numerics contract: the epilogue keeps the fp32 accumulator through gate, bias and residual and rounds ONCE, where the
    eager chain rounds after the GEMM, after the gate multiply and after the add -- closer to exact arithmetic, within two
    ulps of the largest term the eager chain rounds (tests/test_gpu_dit.py).  JENGA_FUSE_GATE=0 is the configuration that
    is bit-comparable with the reference goldens / the oracle for proj, fc2 and linear2."""
    if mask is not None or not x.is_cuda or not FUSE_GATE_RESIDUAL:
        return _capi.gate_residual(res, lin(x), gate, gate2=gate2, mask=mask)
    g = gate.reshape(-1)
    bias = None
    if lin.bias is not None:
        # gate * bias in fp32, handed to the GEMM as a float32 bias vector (JENGA_BIAS_F32): no extra rounding on its way
        # into the fp32 accumulator.  JENGA_GATED_BIAS=bf16 restores round 3's bf16 product (one more rounding)
        bias = lin.bias.float() * g.float() if GATED_BIAS_F32 else lin.bias * g.to(lin.bias.dtype)
    return _capi.linear(x, lin.weight, bias, gate=g, res=res)


def _sp_linear(x, w, b):
    """The plain GEMMs of the sequence-parallel overlap path through jenga_linear (bias epilogue, same arithmetic as
    F.linear): the per-rank shapes (M = S_img / N, the Q|K / V split) then take part in the candidate timing of
    JENGA_GEMM_CANDIDATES and in the choice broadcast across ranks, like the epilogue GEMMs beside them."""
    return _capi.linear(x, w, b)


def _select_top_k(sa_drop_rate, img_block_num):
    return int((1 - sa_drop_rate) * img_block_num)  # Python float truncation: (0.8, 900) -> 179 (models_mul...:242)


class MMDoubleStreamBlock(nn.Module):
    def __init__(self, hidden_size: int, heads_num: int, mlp_width_ratio: float, dtype=None, device=None):
        fk = dict(dtype=dtype, device=device)
        super().__init__()
        self.heads_num = heads_num
        head_dim = hidden_size // heads_num
        mlp_hidden = int(hidden_size * mlp_width_ratio)
        self.img_mod = ModulateDiT(hidden_size, 6, **fk)
        self.img_norm1 = nn.LayerNorm(hidden_size, elementwise_affine=False, eps=1e-6, **fk)
        self.img_attn_qkv = nn.Linear(hidden_size, hidden_size * 3, bias=True, **fk)
        self.img_attn_q_norm = RMSNorm(head_dim, elementwise_affine=True, eps=1e-6, **fk)
        self.img_attn_k_norm = RMSNorm(head_dim, elementwise_affine=True, eps=1e-6, **fk)
        self.img_attn_proj = nn.Linear(hidden_size, hidden_size, bias=True, **fk)
        self.img_norm2 = nn.LayerNorm(hidden_size, elementwise_affine=False, eps=1e-6, **fk)
        self.img_mlp = MLP(hidden_size, mlp_hidden, **fk)
        self.txt_mod = ModulateDiT(hidden_size, 6, **fk)
        self.txt_norm1 = nn.LayerNorm(hidden_size, elementwise_affine=False, eps=1e-6, **fk)
        self.txt_attn_qkv = nn.Linear(hidden_size, hidden_size * 3, bias=True, **fk)
        self.txt_attn_q_norm = RMSNorm(head_dim, elementwise_affine=True, eps=1e-6, **fk)
        self.txt_attn_k_norm = RMSNorm(head_dim, elementwise_affine=True, eps=1e-6, **fk)
        self.txt_attn_proj = nn.Linear(hidden_size, hidden_size, bias=True, **fk)
        self.txt_norm2 = nn.LayerNorm(hidden_size, elementwise_affine=False, eps=1e-6, **fk)
        self.txt_mlp = MLP(hidden_size, mlp_hidden, **fk)
        self.hybrid_seq_parallel_attn = None

    def _attention_unfused(self, img_qkv, txt_qkv, cos, sin, sa_drop_rate, top_k, txt_amp, txt_block_num,
                           p_remain_rates, block_neighbor_list, cu_seqlens_q, cu_seqlens_kv):
        """Single GPU (and a sequence-parallel module that only offers the reference's forward signature)."""
        H = self.heads_num
        B, S_img = img_qkv.shape[:2]
        S_txt = txt_qkv.shape[1]
        # QK-norm + RoPE (+ the block means the selection needs) fused: one kernel per stream handles Q and K, writes
        # straight into the concatenated (image | text) buffers and fills the pooled tensors
        q = torch.empty((B, S_img + S_txt, H, 128), dtype=img_qkv.dtype, device=img_qkv.device)
        k = torch.empty_like(q)
        sparse = (not self.hybrid_seq_parallel_attn) and sa_drop_rate != 0.0 and S_img % 128 == 0 and S_txt % 128 == 0
        pooled = None
        if sparse:      # (sequence parallel: pooling happens after the exchange, on the gathered sequence)
            nimg_, nb_ = S_img // 128, (S_img + S_txt) // 128
            pooled = (torch.empty((B, H, nimg_, 128), dtype=img_qkv.dtype, device=img_qkv.device),
                      torch.empty((B, H, nb_, 128), dtype=img_qkv.dtype, device=img_qkv.device))
        if S_img % 128 == 0 and S_txt % 128 == 0:
            qp, kp = pooled if pooled else (None, None)
            _capi.qk_norm_rope_pool(img_qkv[:, :, 0], img_qkv[:, :, 1], self.img_attn_q_norm.weight,
                                    self.img_attn_k_norm.weight, cos, sin, q[:, :S_img], k[:, :S_img], qpool=qp, kpool=kp)
            _capi.qk_norm_rope_pool(txt_qkv[:, :, 0], txt_qkv[:, :, 1], self.txt_attn_q_norm.weight,
                                    self.txt_attn_k_norm.weight, None, None, q[:, S_img:], k[:, S_img:], qpool=None,
                                    kpool=kp, pool_block0=S_img // 128)
        else:
            _capi.rmsnorm_rope(img_qkv[:, :, 0], self.img_attn_q_norm.weight, cos, sin, out=q[:, :S_img])
            _capi.rmsnorm_rope(img_qkv[:, :, 1], self.img_attn_k_norm.weight, cos, sin, out=k[:, :S_img])
            _capi.rmsnorm_rope(txt_qkv[:, :, 0], self.txt_attn_q_norm.weight, None, None, out=q[:, S_img:])
            _capi.rmsnorm_rope(txt_qkv[:, :, 1], self.txt_attn_k_norm.weight, None, None, out=k[:, S_img:])
        if self.hybrid_seq_parallel_attn:
            # my_parallel_attention's argument convention (attenion.py:159-195) without concatenating V first: the
            # exchange packs the image part and slices the text part separately anyway
            attn = self.hybrid_seq_parallel_attn(
                None, q[:, :S_img], k[:, :S_img], img_qkv[:, :, 2], dropout_p=0.0, causal=False,
                joint_tensor_query=q[:, S_img:], joint_tensor_key=k[:, S_img:], joint_tensor_value=txt_qkv[:, :, 2],
                joint_strategy="rear", top_k=ulysses.get_sequence_parallel_world_size() * top_k,
                cu_seqlens_q=cu_seqlens_q, cu_seqlens_kv=cu_seqlens_kv, text_amp=txt_amp,
                block_neighbor_list=block_neighbor_list, p_remain_rates=p_remain_rates).reshape(B, S_img + S_txt, -1)
        elif sa_drop_rate == 0.0:
            v = torch.cat((img_qkv[:, :, 2], txt_qkv[:, :, 2]), dim=1)
            attn = dense_attention(q, k, v, cu_seqlens_q=cu_seqlens_q, cu_seqlens_kv=cu_seqlens_kv)
        else:
            nb, nimg = (S_img + S_txt) // 128, S_img // 128
            vt = _capi.pack_v(img_qkv[:, :, 2], nimg, dst_block0=0, dst_blocks_total=nb)
            _capi.pack_v(txt_qkv[:, :, 2], S_txt // 128, out=vt, dst_block0=nimg, dst_blocks_total=nb)
            seqlens = cu_seqlens_q[1:2]
            attn = op.attencarve_packed(q, k, vt, top_k, seqlens, txt_block_num, txt_amp, p_remain_rates,
                                        block_neighbor_list, pooled=pooled).view(B, S_img + S_txt, H * 128)
        return attn

    @torch.no_grad()   # inference only, like the reference (hyvideo/inference.py:195 disables grad globally)
    def forward(self, img, txt, vec, cu_seqlens_q=None, cu_seqlens_kv=None, max_seqlen_q=None, max_seqlen_kv=None,
                freqs_cis: tuple = None, sa_drop_rate: float = 0.0, txt_amp: float = 1.0, curve_sel: list = None,
                p_remain_rates: float = 0.5, txt_block_num: int = 2, per_block_token: int = 128,
                token_replace_vec=None, first_frame_mask=None):
        """token_replace_vec / first_frame_mask: the I2V "token_replace" conditioning (hyvideo_i2v/modules/
        models_mul.py:136-320): image tokens with first_frame_mask set are modulated / gated by a second vector."""
        H = self.heads_num
        B, S_img, C = img.shape
        S_txt = txt.shape[1]
        (img_mod1_shift, img_mod1_scale, img_mod1_gate, img_mod2_shift, img_mod2_scale,
         img_mod2_gate) = self.img_mod(vec).chunk(6, dim=-1)
        (txt_mod1_shift, txt_mod1_scale, txt_mod1_gate, txt_mod2_shift, txt_mod2_scale,
         txt_mod2_gate) = self.txt_mod(vec).chunk(6, dim=-1)
        tr = [None] * 6
        if token_replace_vec is not None:
            tr = self.img_mod(token_replace_vec).chunk(6, dim=-1)
        fm = first_frame_mask if token_replace_vec is not None else None
        block_neighbor_list = curve_sel[0][2] if curve_sel is not None else None
        top_k = _select_top_k(sa_drop_rate, S_img // per_block_token)
        cos, sin = freqs_cis
        sp = self.hybrid_seq_parallel_attn
        fused_sp = bool(sp) and hasattr(sp, "begin")
        # LayerNorm + adaLN modulation fused (one pass over HBM instead of three)
        xm = _capi.ln_modulate(img, img_mod1_shift, img_mod1_scale, shift2=tr[0], scale2=tr[1], mask=fm)
        pend = None
        if fused_sp and SP_OVERLAP:
            # sequence parallel with overlap: Q|K GEMM -> prologue (RMSNorm + RoPE + peer-major pack, any shard length)
            # -> Q, K exchange in flight; the V GEMM, its pack and the whole text stream run under it
            C = H * 128
            w, b = self.img_attn_qkv.weight, self.img_attn_qkv.bias
            qk = _sp_linear(xm, w[: 2 * C], None if b is None else b[: 2 * C]).view(B, S_img, 2, H, 128)
            pend = sp.begin(B, S_img, H, S_txt, qk.dtype, qk.device)
            pend.post_qk(qk[:, :, 0], qk[:, :, 1], (self.img_attn_q_norm.weight, self.img_attn_k_norm.weight), (cos, sin))
            pend.post_v(_sp_linear(xm, w[2 * C:], None if b is None else b[2 * C:]).view(B, S_img, H, 128))
        else:
            img_qkv = self.img_attn_qkv(xm).view(B, S_img, 3, H, 128)
            if fused_sp:
                pend = sp.begin(B, S_img, H, S_txt, img_qkv.dtype, img_qkv.device)
                pend.post_qkv(img_qkv[:, :, 0], img_qkv[:, :, 1], img_qkv[:, :, 2],
                              (self.img_attn_q_norm.weight, self.img_attn_k_norm.weight), (cos, sin))
        if (pend is None and ROWOPS_OVERLAP and not sp and img.is_cuda and sa_drop_rate != 0.0 and S_img % 128 == 0
                and S_txt % 128 == 0 and S_txt // 128 == txt_block_num):
            attn = self._attention_overlapped(img_qkv, txt, txt_mod1_shift, txt_mod1_scale, cos, sin, top_k, txt_amp,
                                              txt_block_num, p_remain_rates, block_neighbor_list, cu_seqlens_q)
            return self._after_attention(img, txt, attn, S_img, img_mod1_gate, img_mod2_shift, img_mod2_scale, img_mod2_gate,
                                         txt_mod1_gate, txt_mod2_shift, txt_mod2_scale, txt_mod2_gate, tr, fm)
        txt_qkv = self.txt_attn_qkv(_capi.ln_modulate(txt, txt_mod1_shift, txt_mod1_scale)).view(B, S_txt, 3, H, 128)
        if pend is not None:
            # this rank's head slice of the (replicated) text rows goes straight behind the gathered image rows; pooling
            # happens after the exchange, on the gathered sequence
            pend.put_text(txt_qkv[:, :, 0], txt_qkv[:, :, 1], txt_qkv[:, :, 2],
                          (self.txt_attn_q_norm.weight, self.txt_attn_k_norm.weight))
            attn = pend.finish(top_k=ulysses.get_sequence_parallel_world_size() * top_k, text_amp=txt_amp,
                               block_neighbor_list=block_neighbor_list, p_remain_rates=p_remain_rates,
                               cu_seqlens_q=cu_seqlens_q).reshape(B, S_img + S_txt, -1)
        else:
            attn = self._attention_unfused(img_qkv, txt_qkv, cos, sin, sa_drop_rate, top_k, txt_amp, txt_block_num,
                                           p_remain_rates, block_neighbor_list, cu_seqlens_q, cu_seqlens_kv)
        return self._after_attention(img, txt, attn, S_img, img_mod1_gate, img_mod2_shift, img_mod2_scale, img_mod2_gate,
                                     txt_mod1_gate, txt_mod2_shift, txt_mod2_scale, txt_mod2_gate, tr, fm)

    def _attention_overlapped(self, img_qkv, txt, txt_shift, txt_scale, cos, sin, top_k, txt_amp, txt_block_num,
                              p_remain_rates, block_neighbor_list, cu_seqlens_q):
        """The single-GPU sparse branch of _attention_unfused with the image stream's row kernels (norm + RoPE + pooling of
        115 200 rows, V re-tiling) on the side stream while the text stream's LayerNorm + modulate, QKV GEMM, norm and V
        re-tiling run on the main one -- they write disjoint parts of the same (image | text) buffers.  Same kernels, same
        inputs: bit-identical to the one-stream order."""
        H = self.heads_num
        B, S_img = img_qkv.shape[:2]
        S_txt = txt.shape[1]
        dt, dev = img_qkv.dtype, img_qkv.device
        nimg, nb = S_img // 128, (S_img + S_txt) // 128
        q = torch.empty((B, S_img + S_txt, H, 128), dtype=dt, device=dev)
        k = torch.empty_like(q)
        qp = torch.empty((B, H, nimg, 128), dtype=dt, device=dev)
        kp = torch.empty((B, H, nb, 128), dtype=dt, device=dev)
        vt = torch.empty((B, H, 2 * nb, 128, 64), dtype=dt, device=dev)

        def image_rows():
            _capi.qk_norm_rope_pool(img_qkv[:, :, 0], img_qkv[:, :, 1], self.img_attn_q_norm.weight,
                                    self.img_attn_k_norm.weight, cos, sin, q[:, :S_img], k[:, :S_img], qpool=qp, kpool=kp)
            _capi.pack_v(img_qkv[:, :, 2], nimg, out=vt, dst_block0=0, dst_blocks_total=nb)

        side = _Fork(dev, image_rows)
        txt_qkv = self.txt_attn_qkv(_capi.ln_modulate(txt, txt_shift, txt_scale)).view(B, S_txt, 3, H, 128)
        _capi.qk_norm_rope_pool(txt_qkv[:, :, 0], txt_qkv[:, :, 1], self.txt_attn_q_norm.weight,
                                self.txt_attn_k_norm.weight, None, None, q[:, S_img:], k[:, S_img:], qpool=None,
                                kpool=kp, pool_block0=nimg)
        _capi.pack_v(txt_qkv[:, :, 2], S_txt // 128, out=vt, dst_block0=nimg, dst_blocks_total=nb)
        side.join()
        return op.attencarve_packed(q, k, vt, top_k, cu_seqlens_q[1:2], txt_block_num, txt_amp, p_remain_rates,
                                    block_neighbor_list, pooled=(qp, kp)).view(B, S_img + S_txt, H * 128)

    def _after_attention(self, img, txt, attn, S_img, img_mod1_gate, img_mod2_shift, img_mod2_scale, img_mod2_gate,
                         txt_mod1_gate, txt_mod2_shift, txt_mod2_scale, txt_mod2_gate, tr, fm):
        img_attn, txt_attn = attn[:, :S_img], attn[:, S_img:]
        # gate * proj(attn) + residual in the proj GEMM's epilogue; the MLP input is LayerNorm + modulate in one pass, its
        # fc1 carries the GELU, its fc2 the gate and the residual
        img = linear_gate_residual(self.img_attn_proj, img_attn, img_mod1_gate, img, gate2=tr[2], mask=fm)
        img = linear_gate_residual(self.img_mlp.fc2,
                                   self.img_mlp.hidden(_capi.ln_modulate(img, img_mod2_shift, img_mod2_scale,
                                                                         shift2=tr[3], scale2=tr[4], mask=fm)),
                                   img_mod2_gate, img, gate2=tr[5], mask=fm)
        txt = linear_gate_residual(self.txt_attn_proj, txt_attn, txt_mod1_gate, txt)
        txt = linear_gate_residual(self.txt_mlp.fc2,
                                   self.txt_mlp.hidden(_capi.ln_modulate(txt, txt_mod2_shift, txt_mod2_scale)),
                                   txt_mod2_gate, txt)
        return img, txt


class MMSingleStreamBlock(nn.Module):
    def __init__(self, hidden_size: int, heads_num: int, mlp_width_ratio: float = 4.0, dtype=None, device=None):
        fk = dict(dtype=dtype, device=device)
        super().__init__()
        self.hidden_size = hidden_size
        self.heads_num = heads_num
        head_dim = hidden_size // heads_num
        self.mlp_hidden_dim = int(hidden_size * mlp_width_ratio)
        self.linear1 = nn.Linear(hidden_size, hidden_size * 3 + self.mlp_hidden_dim, **fk)
        self.linear2 = nn.Linear(hidden_size + self.mlp_hidden_dim, hidden_size, **fk)
        self.q_norm = RMSNorm(head_dim, elementwise_affine=True, eps=1e-6, **fk)
        self.k_norm = RMSNorm(head_dim, elementwise_affine=True, eps=1e-6, **fk)
        self.pre_norm = nn.LayerNorm(hidden_size, elementwise_affine=False, eps=1e-6, **fk)
        self.modulation = ModulateDiT(hidden_size, 3, **fk)
        self.hybrid_seq_parallel_attn = None

    def _attention_unfused(self, qkv, S_img, cos, sin, sa_drop_rate, top_k, txt_amp, txt_block_num, p_remain_rates,
                           block_neighbor_list, cu_seqlens_q, cu_seqlens_kv, cat, attn_out):
        H, C = self.heads_num, self.hidden_size
        B, S = qkv.shape[:2]
        sparse = (not self.hybrid_seq_parallel_attn) and sa_drop_rate != 0.0 and S % 128 == 0 and S_img % 128 == 0
        pooled = None
        if S % 128 == 0:   # Q and K in one kernel (RoPE on image tokens only), block means for the selection on the way
            q = torch.empty((B, S, H, 128), dtype=qkv.dtype, device=qkv.device)
            k = torch.empty_like(q)
            if sparse:
                pooled = (torch.empty((B, H, S_img // 128, 128), dtype=qkv.dtype, device=qkv.device),
                          torch.empty((B, H, S // 128, 128), dtype=qkv.dtype, device=qkv.device))
            _capi.qk_norm_rope_pool(qkv[:, :, 0], qkv[:, :, 1], self.q_norm.weight, self.k_norm.weight, cos, sin, q, k,
                                    s_rope=S_img, qpool=pooled[0] if pooled else None,
                                    kpool=pooled[1] if pooled else None)
        else:
            q = _capi.rmsnorm_rope(qkv[:, :, 0], self.q_norm.weight, cos, sin, s_rope=S_img)
            k = _capi.rmsnorm_rope(qkv[:, :, 1], self.k_norm.weight, cos, sin, s_rope=S_img)
        if self.hybrid_seq_parallel_attn:
            attn = my_parallel_attention(self.hybrid_seq_parallel_attn, q, k, qkv[:, :, 2], img_q_len=S_img,
                                         img_kv_len=S_img, cu_seqlens_q=cu_seqlens_q, cu_seqlens_kv=cu_seqlens_kv,
                                         top_k=ulysses.get_sequence_parallel_world_size() * top_k, text_amp=txt_amp,
                                         block_neighbor_list=block_neighbor_list, p_remain_rates=p_remain_rates)
            cat[..., :C] = attn
        elif sa_drop_rate == 0.0:
            cat[..., :C] = dense_attention(q, k, qkv[:, :, 2], cu_seqlens_q=cu_seqlens_q, cu_seqlens_kv=cu_seqlens_kv)
        else:
            vt = _capi.pack_v(qkv[:, :, 2], S // 128)
            op.attencarve_packed(q, k, vt, top_k, cu_seqlens_q[1:2], txt_block_num, txt_amp, p_remain_rates,
                                 block_neighbor_list, out=attn_out, pooled=pooled)

    def _attention_prepare(self, qkv, S_img, cos, sin, top_k, txt_block_num, p_remain_rates, block_neighbor_list):
        """Everything of the single-GPU AttenCarve call in front of the attention launch (the sparse branch of
        _attention_unfused, same kernels in the same order): fused Q / K RMSNorm + RoPE + block means, V re-tiling, block
        selection.  -> (q, k, vt, idx, cnt)"""
        H = self.heads_num
        B, S = qkv.shape[:2]
        q = torch.empty((B, S, H, 128), dtype=qkv.dtype, device=qkv.device)
        k = torch.empty_like(q)
        pooled = (torch.empty((B, H, S_img // 128, 128), dtype=qkv.dtype, device=qkv.device),
                  torch.empty((B, H, S // 128, 128), dtype=qkv.dtype, device=qkv.device))
        _capi.qk_norm_rope_pool(qkv[:, :, 0], qkv[:, :, 1], self.q_norm.weight, self.k_norm.weight, cos, sin, q, k,
                                s_rope=S_img, qpool=pooled[0], kpool=pooled[1])
        vt = _capi.pack_v(qkv[:, :, 2], S // 128)
        _, idx, cnt = op.build_block_index(q, k, top_k, txt_block_num, p_remain_rates, block_neighbor_list, pooled=pooled)
        return q, k, vt, idx, cnt

    @torch.no_grad()
    def forward(self, x, vec, txt_len, cu_seqlens_q=None, cu_seqlens_kv=None, max_seqlen_q=None, max_seqlen_kv=None,
                freqs_cis: Tuple[torch.Tensor, torch.Tensor] = None, sa_drop_rate: float = 0.0, txt_amp: float = 1.0,
                curve_sel: list = None, p_remain_rates: float = 0.5, txt_block_num: int = 2,
                per_block_token: int = 128, token_replace_vec=None, first_frame_mask=None):
        H, C = self.heads_num, self.hidden_size
        B, S, _ = x.shape
        S_img = S - txt_len
        mod_shift, mod_scale, mod_gate = self.modulation(vec).chunk(3, dim=-1)
        tr, fm = [None] * 3, None
        if token_replace_vec is not None:       # I2V token_replace (hyvideo_i2v/modules/models_mul.py:393-506)
            tr = self.modulation(token_replace_vec).chunk(3, dim=-1)
            fm = torch.cat([first_frame_mask.to(torch.uint8),
                            torch.zeros(txt_len, dtype=torch.uint8, device=x.device)])   # text rows: regular set
        # linear1 as two GEMMs over the same input: the QKV half plain, the MLP half with the tanh-GELU in its epilogue
        # and linear2's concat buffer as its (strided) destination -- no separate 5.7 GB activation pass, no copy
        xm = _capi.ln_modulate(x, mod_shift, mod_scale, shift2=tr[0], scale2=tr[1], mask=fm)
        cat = torch.empty((B, S, C + self.mlp_hidden_dim), dtype=x.dtype, device=x.device)
        cos, sin = freqs_cis
        block_neighbor_list = curve_sel[0][2] if curve_sel is not None else None
        top_k = _select_top_k(sa_drop_rate, S_img // per_block_token)
        # attention writes its [B,S,H*128] output straight into the left part of the concat buffer
        attn_out = cat[..., :C].unflatten(-1, (H, 128))
        sp = self.hybrid_seq_parallel_attn
        fused_sp = bool(sp) and hasattr(sp, "begin")
        w1, b1 = self.linear1.weight, self.linear1.bias

        def mlp_half(c0, c1):      # columns [c0, c1) of the MLP half: GEMM + tanh-GELU epilogue into the concat buffer
            if c1 > c0:
                _capi.linear(xm, w1[3 * C + c0: 3 * C + c1], None if b1 is None else b1[3 * C + c0: 3 * C + c1],
                             act=_capi.ACT_GELU_TANH, out=cat[..., C + c0: C + c1])

        if fused_sp and SPLIT_LINEAR1:
            # sequence parallel: QKV half -> fused prologue (image rows -> peer-major send buffers, this rank's head slice
            # of the text rows -> in place) -> Q, K, V in flight; the MLP half runs under the exchange, its tail under
            # the O exchange; the unpack kernels write into the concat buffer
            w = (self.q_norm.weight, self.k_norm.weight)
            pend = sp.begin(B, S_img, H, S - S_img, xm.dtype, xm.device)
            if SP_OVERLAP:      # Q|K GEMM -> Q, K exchange posted; the V GEMM already runs under it
                qk = _sp_linear(xm, w1[: 2 * C], None if b1 is None else b1[: 2 * C]).unflatten(-1, (2, H, 128))
                pend.post_qk(qk[:, :S_img, 0], qk[:, :S_img, 1], w, (cos, sin))
                v = _sp_linear(xm, w1[2 * C: 3 * C], None if b1 is None else b1[2 * C: 3 * C]).unflatten(-1, (H, 128))
                pend.post_v(v[:, :S_img])
                pend.put_text(qk[:, S_img:, 0], qk[:, S_img:, 1], v[:, S_img:], w)
            else:
                qkv = F.linear(xm, w1[: 3 * C], None if b1 is None else b1[: 3 * C]).unflatten(-1, (3, H, 128))
                pend.post_qkv(qkv[:, :S_img, 0], qkv[:, :S_img, 1], qkv[:, :S_img, 2], w, (cos, sin))
                pend.put_text(qkv[:, S_img:, 0], qkv[:, S_img:, 1], qkv[:, S_img:, 2], w)
            Mh = self.mlp_hidden_dim
            tail = (int(Mh * SP_MLP_TAIL) // 256) * 256 if SP_OVERLAP else 0
            mlp_half(0, Mh - tail)
            pend.finish(top_k=ulysses.get_sequence_parallel_world_size() * top_k, text_amp=txt_amp,
                        block_neighbor_list=block_neighbor_list, p_remain_rates=p_remain_rates,
                        cu_seqlens_q=cu_seqlens_q, out=attn_out, while_out=lambda: mlp_half(Mh - tail, Mh))
            return linear_gate_residual(self.linear2, cat, mod_gate, x, gate2=tr[2], mask=fm)
        # linear1 as two GEMMs over the same input: the QKV half plain, the MLP half with the tanh-GELU in its epilogue
        # and linear2's concat buffer as its (strided) destination -- no separate 5.7 GB activation pass, no copy
        if SPLIT_LINEAR1:
            qkv = F.linear(xm, w1[: 3 * C], None if b1 is None else b1[: 3 * C]).unflatten(-1, (3, H, 128))
            if (ROWOPS_OVERLAP and not sp and x.is_cuda and sa_drop_rate != 0.0 and S % 128 == 0 and S_img % 128 == 0
                    and S // 128 > txt_block_num):
                # the row kernels on the side stream, the MLP half beside them, then the attention launch
                prep = _Fork(x.device, lambda: self._attention_prepare(qkv, S_img, cos, sin, top_k, txt_block_num,
                                                                       p_remain_rates, block_neighbor_list))
                mlp_half(0, self.mlp_hidden_dim)
                q, k, vt, idx, cnt = prep.join()
                nimg = S // 128 - txt_block_num
                _capi.bsattn_fwd(q, k, vt, cu_seqlens_q[1:2], idx, cnt, nimg, 128 ** -0.5, txt_amp, nimg, out=attn_out)
                return linear_gate_residual(self.linear2, cat, mod_gate, x, gate2=tr[2], mask=fm)
            mlp_half(0, self.mlp_hidden_dim)
        else:       # one GEMM, then the activation as a pass of its own (strided source and destination)
            lin1 = self.linear1(xm)
            qkv = lin1[..., : 3 * C].unflatten(-1, (3, H, 128))
            _capi.gelu_tanh(lin1[..., 3 * C:], out=cat[..., C:])
        if fused_sp:
            w = (self.q_norm.weight, self.k_norm.weight)
            sp.forward_qkv(tuple(qkv[:, :S_img, i] for i in range(3)), tuple(qkv[:, S_img:, i] for i in range(3)), w, w,
                           (cos, sin), top_k=ulysses.get_sequence_parallel_world_size() * top_k, text_amp=txt_amp,
                           block_neighbor_list=block_neighbor_list, p_remain_rates=p_remain_rates,
                           cu_seqlens_q=cu_seqlens_q, out=attn_out)
        else:
            self._attention_unfused(qkv, S_img, cos, sin, sa_drop_rate, top_k, txt_amp, txt_block_num, p_remain_rates,
                                    block_neighbor_list, cu_seqlens_q, cu_seqlens_kv, cat, attn_out)
        return linear_gate_residual(self.linear2, cat, mod_gate, x, gate2=tr[2], mask=fm)


class JengaHYVideoDiT(nn.Module):
    """Synthetic-weight HunyuanVideo DiT driven the way jenga_hyvideo.py drives the real one: Jenga state lives in
    attributes set by the caller (jenga_hyvideo.py:275-287: enable_skip, cnt, num_steps, curve_sel, sa_drop_rate,
    text_amp, p_remain_rates, linear_to_hilbert, hilbert_order, start_stage, previous_residual)."""

    def __init__(self, config="HYVideo-T/2-cfgdistill", in_channels=16, out_channels=16, patch_size=(1, 2, 2),
                 text_states_dim=4096, text_states_dim_2=768, depth_double=None, depth_single=None, dtype=torch.bfloat16,
                 device=None):
        super().__init__()
        cfg = HUNYUAN_VIDEO_CONFIG[config]
        self.hidden_size, self.heads_num = cfg["hidden_size"], cfg["heads_num"]
        self.rope_dim_list = cfg["rope_dim_list"]
        self.patch_size = list(patch_size)
        self.out_channels = out_channels
        fk = dict(dtype=dtype, device=device)
        pdim = in_channels * math.prod(patch_size)
        self.img_in = nn.Linear(pdim, self.hidden_size, **fk)             # PatchEmbed (conv3d, stride = kernel) as a GEMM
        self.txt_in = nn.Linear(text_states_dim, self.hidden_size, **fk)  # stands in for the token refiner
        self.time_in = nn.Linear(256, self.hidden_size, **fk)
        self.vector_in = nn.Linear(text_states_dim_2, self.hidden_size, **fk)
        self.guidance_in = nn.Linear(256, self.hidden_size, **fk)
        nd = cfg["mm_double_blocks_depth"] if depth_double is None else depth_double
        ns = cfg["mm_single_blocks_depth"] if depth_single is None else depth_single
        self.double_blocks = nn.ModuleList(
            [MMDoubleStreamBlock(self.hidden_size, self.heads_num, cfg["mlp_width_ratio"], **fk) for _ in range(nd)])
        self.single_blocks = nn.ModuleList(
            [MMSingleStreamBlock(self.hidden_size, self.heads_num, cfg["mlp_width_ratio"], **fk) for _ in range(ns)])
        self.final_norm = nn.LayerNorm(self.hidden_size, elementwise_affine=False, eps=1e-6, **fk)
        self.final_mod = nn.Linear(self.hidden_size, 2 * self.hidden_size, **fk)
        self.final_linear = nn.Linear(self.hidden_size, math.prod(patch_size) * out_channels, **fk)
        # Jenga state
        self.enable_skip = True
        self.cnt = 0
        self.num_steps = 50
        self.start_stage = False
        self.previous_residual = None
        self.curve_sel = None
        self.linear_to_hilbert = None
        self.hilbert_order = None
        self.sa_drop_rate = 0.0
        self.text_amp = 0.0
        self.p_remain_rates = 0.3
        # HunyuanVideo-I2V (jenga_hyi2v.py:79-92, 124-130): "token_replace" conditions the first latent frame's tokens
        # with the modulation of timestep 0; None = text-to-video
        self.i2v_condition_type = None
        self.sp_check_agreement = os.environ.get("JENGA_SP_CHECK", "0") == "1"

    @torch.no_grad()
    def init_synthetic_weights(self, std=0.02, seed=0):
        g = torch.Generator(device=self.img_in.weight.device).manual_seed(seed)
        for name, p in self.named_parameters():
            if name.endswith("norm.weight") or "_norm.weight" in name:
                p.copy_((1 + 0.1 * torch.randn(p.shape, generator=g, device=p.device)).to(p.dtype))
            elif p.dim() >= 2:
                p.normal_(0.0, std, generator=g)
            else:
                p.normal_(0.0, std, generator=g)   # biases and modulation are non-zero so that gates are exercised
        return self

    # ---- static geometry for one resolution stage (jenga_hyvideo.py:43-58 + pipeline...prores.py:238-284) ----
    def set_stage(self, latent_thw, device):
        t, h, w = latent_thw
        pt, ph, pw = self.patch_size
        tt, th, tw = t // pt, h // ph, w // pw
        l2h, h2l = G.gilbert_mapping(tt, th, tw, as_tensor=True, device=device)
        nbm = G.gilbert_block_neighbor_mapping(tt, th, tw, as_tensor=True, device=device)
        self.curve_sel = [[l2h, h2l, nbm]]
        self.linear_to_hilbert, self.hilbert_order = l2h, h2l
        cos, sin = get_nd_rotary_pos_embed(self.rope_dim_list, [tt, th, tw], theta=256, use_real=True,
                                           theta_rescale_factor=1)
        return cos.to(device), sin.to(device)

    def patchify(self, x):
        B, C, T, Hh, W = x.shape
        pt, ph, pw = self.patch_size
        x = x.view(B, C, T // pt, pt, Hh // ph, ph, W // pw, pw).permute(0, 2, 4, 6, 1, 3, 5, 7)
        return x.reshape(B, (T // pt) * (Hh // ph) * (W // pw), C * pt * ph * pw)

    def unpatchify(self, x, t, h, w):
        c = self.out_channels
        pt, ph, pw = self.patch_size
        x = x.reshape(x.shape[0], t, h, w, c, pt, ph, pw)
        x = torch.einsum("nthwcopq->nctohpwq", x)
        return x.reshape(x.shape[0], c, t * pt, h * ph, w * pw)

    @staticmethod
    def _sinusoid(t, dim=256):
        half = dim // 2
        f = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
        a = t.float()[:, None] * f[None]
        return torch.cat([a.cos(), a.sin()], dim=-1)

    @torch.no_grad()
    def forward(self, x, t, text_states=None, text_mask=None, text_states_2=None, freqs_cos=None, freqs_sin=None,
                guidance=None, return_dict=True):
        """Counterpart of ra_forward (single GPU) and of new_forward+transformer_sub_forward (sequence parallel: the
        reordered image tokens and RoPE rows are split into N contiguous shards, jenga_hyvideo_multigpu.py:168-177)."""
        _, _, ot, oh, ow = x.shape
        tt, th, tw = ot // self.patch_size[0], oh // self.patch_size[1], ow // self.patch_size[2]
        dt = self.img_in.weight.dtype
        vec = self.time_in(self._sinusoid(t).to(dt)) + self.vector_in(text_states_2.to(dt))
        vec = vec + self.guidance_in(self._sinusoid(guidance).to(dt))
        txt = self.txt_in(text_states.to(dt))
        patches = self.patchify(x).to(dt)
        txt_seq_len, img_seq_len = txt.shape[1], patches.shape[1]
        sp = self.double_blocks[0].hybrid_seq_parallel_attn if len(self.double_blocks) else None
        order = self.hilbert_order
        n, r = 1, 0
        if sp:
            # sequence parallel: this rank embeds and carries ONLY its contiguous shard of the curve-ordered tokens
            # (the reference embeds everything on every rank and chunks afterwards, jenga_hyvideo_multigpu.py:168-177;
            # token-wise layers commute with the gather, so gathering the shard's rows first is the same arithmetic)
            n, r = ulysses.get_sequence_parallel_world_size(), ulysses.get_sequence_parallel_rank()
            assert img_seq_len % n == 0, f"cannot split {img_seq_len} image tokens over {n} ranks"
            order = torch.chunk(order, n, dim=0)[r].contiguous()
        # Hilbert gather (K10): image patches and RoPE rows into curve order (only this rank's rows)
        img = self.img_in(_capi.gather_rows(patches, order))
        freqs_cos = _capi.gather_rows(freqs_cos.unsqueeze(0), order)[0]
        freqs_sin = _capi.gather_rows(freqs_sin.unsqueeze(0), order)[0]
        token_replace_vec = first_frame_mask = None
        if self.i2v_condition_type == "token_replace":
            token_replace_vec = self.time_in(self._sinusoid(torch.zeros_like(t)).to(dt)) + self.vector_in(text_states_2.to(dt))
            first_frame_mask = (order < th * tw)                       # == mask[:th*tw] = 1 gathered into curve order
        elif self.i2v_condition_type is not None:
            raise ValueError(f"unsupported i2v_condition_type {self.i2v_condition_type!r}")
        if txt_seq_len % 128:
            raise ValueError("the text length must be a multiple of 128 (2 blocks for T2V, 4 for I2V)")
        loc_len = img.shape[1]
        cu_seqlens_q = get_cu_seqlens(text_mask, loc_len)
        cu_seqlens_kv = cu_seqlens_q
        max_seqlen_q = max_seqlen_kv = loc_len + txt_seq_len
        should_calc = (not self.enable_skip) or (self.cnt in NON_SKIP_STEPS) or self.start_stage
        self.start_stage = False
        if sp and self.enable_skip and torch.distributed.is_initialized() and self.sp_check_agreement:
            # C4 (jenga_hyvideo_multigpu.py:233-236): the reference all-reduces the skip flag every step and reads it
            # back on the host.  Every rank holds the same step counter, so the decision is already identical; the
            # all-reduce (and its host synchronisation) is kept as a debug assertion only (JENGA_SP_CHECK=1).
            flag = torch.tensor([1 if should_calc else 0], device=img.device)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN, group=ulysses.get_sp_group().group)
            assert bool(flag.item()) == should_calc, "ranks disagree on the step-skip decision"
        if not should_calc:
            img = img + self.previous_residual
        else:
            ori_img = img
            freqs = (freqs_cos, freqs_sin)
            for block in self.double_blocks:
                img, txt = block(img, txt, vec, cu_seqlens_q, cu_seqlens_kv, max_seqlen_q, max_seqlen_kv, freqs,
                                 self.sa_drop_rate, self.text_amp, self.curve_sel, self.p_remain_rates,
                                 txt_block_num=txt_seq_len // 128, token_replace_vec=token_replace_vec,
                                 first_frame_mask=first_frame_mask)
            xcat = torch.cat((img, txt), 1)
            for block in self.single_blocks:
                xcat = block(xcat, vec, txt_seq_len, cu_seqlens_q, cu_seqlens_kv, max_seqlen_q, max_seqlen_kv, freqs,
                             self.sa_drop_rate, self.text_amp, self.curve_sel, self.p_remain_rates,
                             txt_block_num=txt_seq_len // 128, token_replace_vec=token_replace_vec,
                             first_frame_mask=first_frame_mask)
            img = xcat[:, :loc_len]
            if self.enable_skip:
                self.previous_residual = img - ori_img
        self.cnt += 1
        if self.cnt == self.num_steps:
            self.cnt = 0
        # final layer on the LOCAL shard (token-wise), then gather the 64-channel patches instead of the 3072-channel
        # hidden states (C3: 48x fewer bytes than the reference's all_gather of img, jenga_hyvideo_multigpu.py:193)
        shift, scale = self.final_mod(F.silu(vec)).chunk(2, dim=1)
        img = self.final_linear(modulate(self.final_norm(img), shift, scale))
        if sp:
            img = ulysses.get_sp_group().all_gather(img.contiguous(), dim=1)
        img = _capi.gather_rows(img.contiguous(), self.linear_to_hilbert)            # Hilbert scatter (K10)
        img = self.unpatchify(img, tt, th, tw)
        return {"x": img} if return_dict else img
