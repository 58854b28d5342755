"""Ulysses sequence parallelism for AttenCarve over RCCL (torch.distributed backend "nccl" on ROCm) / gloo on CPU
for tests.  Counterpart of hyvideo/modules/xdit_ring_atten.py:22-222 (xFuserLongContextAttention.forward) and of the
xfuser group accessors the driver uses (jenga_hyvideo_multigpu.py:168-177,193).

What the reference does per layer: 4 all-to-alls in (Q, K, V, text-Q), local block_sparse_attention on H/N heads over
the whole sequence, 2 all-to-alls out (image O, and text O repeated N times = an all-gather in disguise).
What this does: three all_to_all_single calls for Q, K, V whose send buffers are packed peer-major by a HIP kernel and
whose receive buffers ARE the prefix of the [S_img + S_txt] tensors the attention kernel reads (zero unpack copies), no
exchange at all for text Q/K/V (text is replicated: every rank slices its own heads), one all_to_all_single for image
O straight out of the attention output (already peer-major), one all_gather over heads for text O.  xGMI is a full
point-to-point mesh, so the N-1 peer messages of an all-to-all (11 MB each at N=8) ride separate links concurrently.

The arithmetic contract (the part parity tests pin): rank-major sequence concatenation, contiguous head slices
[r*H/N, (r+1)*H/N), top_k passed through unchanged (the caller already multiplied it by N, models_mul...:249-251),
cu_seqlens rebuilt as [0, n_valid_text + S_img, S] (:183-184).
"""
import torch
import torch.distributed as dist

from .. import _capi
from . import attention_block_sparse as _op

_SP_GROUP = None


class _SPGroup:
    """Minimal stand-in for xfuser's group object: .all_gather(x, dim) as used at jenga_hyvideo_multigpu.py:193."""

    def __init__(self, group):
        self.group = group

    def all_gather(self, x, dim=0):
        n = dist.get_world_size(self.group)
        if n == 1:
            return x
        parts = [torch.empty_like(x) for _ in range(n)]
        dist.all_gather(parts, x.contiguous(), group=self.group)
        return torch.cat(parts, dim=dim)


def init_sequence_parallel(group=None):
    global _SP_GROUP
    _SP_GROUP = _SPGroup(group if group is not None else dist.group.WORLD)
    return _SP_GROUP


def get_sp_group():
    if _SP_GROUP is None:
        raise RuntimeError("call jenga_amd.modules.ulysses.init_sequence_parallel() first")
    return _SP_GROUP


def get_sequence_parallel_world_size():
    return dist.get_world_size(get_sp_group().group) if (_SP_GROUP and dist.is_initialized()) else 1


def get_sequence_parallel_rank():
    return dist.get_rank(get_sp_group().group) if (_SP_GROUP and dist.is_initialized()) else 0


def _pack_heads(t, N):
    """[B,S_loc,H,D] -> peer-major [N,B,S_loc,H/N,D]: the HIP re-tiling kernel (device tensors only)."""
    return _capi.ulysses_pack_heads(t if t.stride(-1) == 1 else t.contiguous(), N)


def _unpack_heads(recv, N, out):
    """peer-major [N,B,S_loc,H/N,D] -> out [B,S_loc,H,D] (may be a strided view): HIP kernel (device tensors only)."""
    return _capi.ulysses_unpack_heads(recv, N, out=out)


def _hip_attention(q_all, k_all, v_all, top_k, seqlens, text_blocks, text_amp, p, neighbors):
    """Default local attention: the HIP AttenCarve core on this rank's heads over the whole sequence."""
    nb = q_all.shape[1] // 128
    vt = _capi.pack_v(v_all, nb)
    return _op.attencarve_packed(q_all, k_all, vt, top_k, seqlens, text_blocks, text_amp, p, neighbors)


class UlyssesAttenCarve(torch.nn.Module):
    """Callable with the signature of xFuserLongContextAttention.forward (xdit_ring_atten.py:61-85); assign an
    instance to `block.hybrid_seq_parallel_attn` exactly as jenga_hyvideo_multigpu.py:181-182 does.

    The three local steps are injectable so that the world_size > 1 exchange logic can be exercised on CPU tensors over
    gloo against the oracle (tests/test_ulysses_gloo.py supplies oracle stand-ins); the defaults are the HIP kernels and
    raise on CPU tensors -- there is no CPU path in the product:
      attn_fn(q_all, k_all, v_all, top_k, seqlens, text_blocks, text_amp, p, neighbors) -> [1,S,H/N,D]
      pack_fn(t [B,S_loc,H,D], N) -> [N,B,S_loc,H/N,D];  unpack_fn(recv [N,B,S_loc,H/N,D], N, out [B,S_loc,H,D])"""

    def __init__(self, group=None, attn_fn=None, pack_fn=None, unpack_fn=None):
        super().__init__()
        self.group = group
        self.attn_fn = attn_fn or _hip_attention
        self.pack_fn = pack_fn or _pack_heads
        self.unpack_fn = unpack_fn or _unpack_heads

    def _pg(self):
        return self.group if self.group is not None else get_sp_group().group

    @torch.no_grad()
    def forward(self, attn, query, key, value, *, joint_tensor_query=None, joint_tensor_key=None,
                joint_tensor_value=None, dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1),
                alibi_slopes=None, deterministic=False, return_attn_probs=False, joint_strategy="none", top_k=0,
                text_amp=0.0, block_neighbor_list=None, p_remain_rates=0.0, cu_seqlens_q=None, cu_seqlens_kv=None):
        if joint_strategy != "rear" or joint_tensor_query is None or joint_tensor_key is None \
                or joint_tensor_value is None:
            raise ValueError("jenga_amd Ulysses: only joint_strategy='rear' with text q/k/v (the Jenga call) is supported")
        pg = self._pg()
        N, r = dist.get_world_size(pg), dist.get_rank(pg)
        B, S_loc, H, D = query.shape
        if B != 1:
            raise ValueError("jenga_amd Ulysses: batch must be 1")
        if H % N:
            raise ValueError(f"heads ({H}) must be divisible by the sequence-parallel degree ({N})")
        Hn = H // N
        S_txt = joint_tensor_query.shape[1]
        S_img = S_loc * N
        S = S_img + S_txt
        if S_img % 128 or S_txt % 128:
            raise ValueError("gathered image length and text length must be multiples of 128")
        dev, dt = query.device, query.dtype
        hs = slice(r * Hn, (r + 1) * Hn)
        # ---- exchange in: scatter heads / gather sequence.  Each all-to-all lands in the PREFIX of the buffer the
        #      attention kernel reads (rank-major sequence == contiguous for B == 1): no unpack, no torch.cat.
        gathered = []
        for t, joint in ((query, joint_tensor_query), (key, joint_tensor_key), (value, joint_tensor_value)):
            full = torch.empty((B, S, Hn, D), dtype=dt, device=dev)
            dist.all_to_all_single(full[0, :S_img].view(N, S_loc, Hn, D), self.pack_fn(t, N).view(N, S_loc, Hn, D),
                                   group=pg)
            full[:, S_img:] = joint[:, :, hs]      # text is replicated on every rank: slice my heads, no exchange
            gathered.append(full)
        q_all, k_all, v_all = gathered
        # cu_seqlens = [0, n_valid_text + S_img, S] (xdit_ring_atten.py:105,183-184) -- stays on the device
        seqlens = (cu_seqlens_q[1:2].to(torch.int64) - S_loc + S_img).to(device=dev, dtype=torch.int32)
        out = self.attn_fn(q_all, k_all, v_all, top_k, seqlens, S_txt // 128, text_amp, p_remain_rates,
                           block_neighbor_list)
        # ---- exchange out: image rows (already peer-major: chunk p = rank p's tokens) back to sequence shards;
        #      text rows gathered over heads (the reference repeats them N times and all-to-alls, :206-217)
        o_img = out[0, :S_img].reshape(N, S_loc, Hn, D)
        if not o_img.is_contiguous():
            o_img = o_img.contiguous()
        o_recv = torch.empty((N, S_loc, Hn, D), dtype=dt, device=dev)
        dist.all_to_all_single(o_recv, o_img, group=pg)
        result = torch.empty((B, S_loc + S_txt, H, D), dtype=dt, device=dev)
        self.unpack_fn(o_recv.view(N, B, S_loc, Hn, D), N, result[:, :S_loc])
        txt_parts = [torch.empty((B, S_txt, Hn, D), dtype=dt, device=dev) for _ in range(N)]
        dist.all_gather(txt_parts, out[:, S_img:].contiguous(), group=pg)
        result[:, S_loc:] = torch.cat(txt_parts, dim=2)
        return result


# name the reference uses (jenga_hyvideo_multigpu.py:181)
xFuserLongContextAttention = UlyssesAttenCarve
