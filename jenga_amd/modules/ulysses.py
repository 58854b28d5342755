"""Ulysses sequence parallelism for AttenCarve over RCCL (torch.distributed backend "nccl" on ROCm) / gloo on CPU
for tests.  Counterpart of hyvideo/modules/xdit_ring_atten.py:22-222 (xFuserLongContextAttention.forward) and of the
xfuser group accessors the driver uses (jenga_hyvideo_multigpu.py:168-177,193).

What the reference does per layer: 4 all-to-alls in (Q, K, V, text-Q), local block_sparse_attention on H/N heads over
the whole sequence, 2 all-to-alls out (image O, and text O repeated N times = an all-gather in disguise), all on one
stream, each followed by a permute+contiguous copy.

What this does per layer ("1 fused QKV all-to-all + 1 O all-to-all + 1 small text all-gather"):
  in : Q, K, V of the local sequence shard are packed peer-major by a HIP kernel and leave in ONE grouped exchange
       (one RCCL group = one communication kernel: N-1 sends + N-1 receives per tensor, posted together).  Every
       incoming message lands directly in the prefix of the [S_img + S_txt] tensor the attention kernel reads
       (rank-major sequence == contiguous for B == 1): no unpack pass, no torch.cat.  The exchange runs on RCCL's own
       stream; the compute stream meanwhile copies the replicated text rows (each rank slices its own heads, nothing
       is exchanged for text), waits for Q and K only, pools and selects blocks, and waits for V just before the V
       re-tiling -- the V transfer overlaps the selection.  (JENGA_ULYSSES_EXCHANGE=a2a: three all_to_all_single calls
       instead, issued back to back and waited for the same way.)
  out: one all_to_all_single for image O straight out of the attention output (already peer-major), one
       all_gather_into_tensor over heads for text O, both in flight together, one HIP unpack kernel each.
xGMI is a full point-to-point mesh: the N-1 peer messages of an exchange (11 MB each at N = 8) ride separate links.

The arithmetic contract (the part parity tests pin): rank-major sequence concatenation, contiguous head slices
[r*H/N, (r+1)*H/N), top_k passed through unchanged (the caller already multiplied it by N, models_mul...:249-251),
cu_seqlens rebuilt as [0, n_valid_text + S_img, S] (:183-184).

Round 4: the call is a PENDING object (`UlyssesAttenCarve.begin()` -> `PendingAttenCarve`): the caller posts each
exchange as soon as its operand exists (`post_qk`, `post_v` / `post_qkv`, `put_text` for the local text slice) and puts
its own independent GEMMs between posting and `finish()`; `finish(while_out=...)` runs the caller's work under the O
exchange.  The reference issues everything on one stream (xdit_ring_atten.py:118-131, 212-217).  `forward` (reference
signature) and `forward_qkv` are begin + post + finish in one call.

Round 5: (a) `JENGA_ULYSSES_PIPELINE=1` / `UlyssesAttenCarve(pipeline=True)`: the rank's H/N heads are exchanged and attended
ONE HEAD AT A TIME -- head g's attention runs while head g+1's Q, K, V are still in flight and head g's O exchange rides
under head g+1's attention; the fallback DESIGN.md section 6 names for a fabric that sustains less than the exchange needs.
Selection and attention are per head, so the result is bit-identical to the unpipelined call (tests/test_gpu_ulysses.py);
the price is H/N attention launches of one head each (launch tails), which is why it is opt-in.  (b) every exchange object
counts the bytes that leave the rank (`bytes_out`, `calls`): bench.py's `roofline_xgmi` record reads them.

The exchange object is injectable (`exchange=`): tests drive N simulated ranks in one process on one GPU with an exchange
that really permutes the chunks (tests/test_gpu_ulysses.py, test_gpu_sp_dit.py), bench.py --simulate-ranks replays the
transfers as side-stream delays, the world_size-2 gloo test and tests/test_gpu_rccl.py exercise the collectives themselves.
"""
import os
import threading

import torch
import torch.distributed as dist

from .. import _capi
from . import attention_block_sparse as _op

_SP_GROUP = None
_TLS = threading.local()


class _SPGroup:
    """Minimal stand-in for xfuser's group object: .all_gather(x, dim) as used at jenga_hyvideo_multigpu.py:193."""

    def __init__(self, group):
        self.group = group

    def size(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def rank(self):
        return dist.get_rank(self.group) if dist.is_initialized() else 0

    def all_gather(self, x, dim=0):
        n = self.size()
        if n == 1:
            return x
        x = x.contiguous()
        out = torch.empty((n,) + tuple(x.shape), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out.view((-1,) + tuple(x.shape[1:])), x, group=self.group)
        return torch.cat(list(out.unbind(0)), dim=dim) if dim != 0 else out.reshape((-1,) + tuple(x.shape[1:]))


def init_sequence_parallel(group=None):
    global _SP_GROUP
    _SP_GROUP = _SPGroup(group if group is not None else dist.group.WORLD)
    return _SP_GROUP


def init_distributed_environment(rank=None, world_size=None, **_ignored):
    """xfuser.core.distributed.init_distributed_environment as hyvideo/inference.py:176 calls it, after
    dist.init_process_group("nccl"): nothing to set up beyond checking that the process group is the one described."""
    if not dist.is_initialized():
        raise RuntimeError("call torch.distributed.init_process_group('nccl') first (hyvideo/inference.py:171)")
    if rank is not None and rank != dist.get_rank() or world_size is not None and world_size != dist.get_world_size():
        raise ValueError("init_distributed_environment: rank / world_size disagree with the process group")


def initialize_model_parallel(sequence_parallel_degree=None, ring_degree=1, ulysses_degree=None, **_ignored):
    """xfuser.core.distributed.initialize_model_parallel as hyvideo/inference.py:178-182 calls it: Jenga's sequence-parallel
    path is Ulysses only (the reference's ring attention is dead code for it): the whole world is one
    sequence-parallel group."""
    if ring_degree not in (None, 1):
        raise ValueError("jenga_amd: ring_degree must be 1 (Ulysses sequence parallelism only)")
    n = dist.get_world_size()
    for name, v in (("sequence_parallel_degree", sequence_parallel_degree), ("ulysses_degree", ulysses_degree)):
        if v is not None and v != n:
            raise ValueError(f"jenga_amd: {name}={v} must equal the world size {n}")
    return init_sequence_parallel()


def set_thread_sp_group(group_like):
    """Per-thread override of the sequence-parallel group: an object with .size(), .rank(), .all_gather(x, dim).
    tests/test_gpu_sp_dit.py runs N simulated ranks as N threads of one process, each with its own in-process group;
    None removes the override.  Production code never calls this."""
    _TLS.group = group_like


def get_sp_group():
    g = getattr(_TLS, "group", None)
    if g is not None:
        return g
    if _SP_GROUP is None:
        raise RuntimeError("call jenga_amd.modules.ulysses.init_sequence_parallel() first")
    return _SP_GROUP


def get_sequence_parallel_world_size():
    g = getattr(_TLS, "group", None)
    if g is not None:
        return g.size()
    return _SP_GROUP.size() if _SP_GROUP else 1


def get_sequence_parallel_rank():
    g = getattr(_TLS, "group", None)
    if g is not None:
        return g.rank()
    return _SP_GROUP.rank() if _SP_GROUP else 0


def _pack_heads(t, N):
    """[B,S_loc,H,D] -> peer-major [N,B,S_loc,H/N,D]: the HIP re-tiling kernel (device tensors only)."""
    return _capi.ulysses_pack_heads(t if t.stride(-1) == 1 else t.contiguous(), N)


def _unpack_heads(recv, N, out):
    """peer-major [N,B,S_loc,H/N,D] -> out [B,S_loc,H,D] (may be a strided view): HIP kernel (device tensors only)."""
    return _capi.ulysses_unpack_heads(recv, N, out=out)


def _hip_select(q_all, k_all, top_k, text_blocks, p, neighbors):
    nb = q_all.shape[1] // 128
    if nb - text_blocks <= 0:
        return None, None
    _, idx, cnt = _op.build_block_index(q_all, k_all, top_k, text_blocks, p, neighbors)
    return idx, cnt


def _hip_attend_dense(q_all, k_all, v_all, seqlens):
    """Every kv block kept for EVERY query block, the kv-length mask on for the text rows too: what LongContextAttention
    computes over image + valid text (parallel_attention, attenion.py:198-221).  Same kernel, same lists as
    jenga_amd.modules.attention.attention (the single-rank dense front-end)."""
    from .attention import _dense_lists, padding_segment
    B, S, Hn, D = q_all.shape
    nb = S // 128
    idx, cnt = _dense_lists(q_all.device, B, Hn, nb)
    o = _capi.bsattn_fwd(q_all, k_all, _capi.pack_v(v_all, nb), seqlens, idx, cnt, nb, D ** -0.5, 0.0, nb)
    return padding_segment(q_all, k_all, v_all, seqlens, o)      # the text-padding rows among themselves (:222-247)


def _hip_attend(q_all, k_all, v_all, idx, cnt, seqlens, text_blocks, text_amp):
    nb = q_all.shape[1] // 128
    vt = _capi.pack_v(v_all, nb)
    return _capi.bsattn_fwd(q_all, k_all, vt, seqlens, idx, cnt, nb - text_blocks, q_all.shape[-1] ** -0.5, text_amp,
                            nb - text_blocks)


class _Done:
    def wait(self):
        return True


class DistExchange:
    """The two exchange steps on a torch.distributed process group.  Every method returns an object with .wait():
    on RCCL the transfer runs on the process group's own stream and .wait() only makes the CURRENT stream wait for it
    (no host synchronisation); on gloo .wait() blocks the host."""

    def __init__(self, group, mode=None):
        self.group = group
        self.mode = mode or os.environ.get("JENGA_ULYSSES_EXCHANGE", "p2p")
        if self.mode not in ("p2p", "a2a"):
            raise ValueError("JENGA_ULYSSES_EXCHANGE must be 'p2p' (one grouped exchange) or 'a2a'")
        self.bytes_out = 0       # bytes that left this rank through this object (algorithmic: (N-1)/N of every send buffer)
        self.calls = 0

    def size(self):
        return dist.get_world_size(self.group)

    def rank(self):
        return dist.get_rank(self.group)

    def all_to_all(self, recvs, sends):
        """recvs[i], sends[i]: [N, ...] contiguous, chunk p of sends[i] goes to rank p and chunk p of recvs[i] comes
        from rank p -- for ALL i in one grouped exchange ("p2p") or one all_to_all_single per tensor ("a2a")."""
        N, r = self.size(), self.rank()
        self.calls += 1
        self.bytes_out += sum(sd.numel() * sd.element_size() for sd in sends) * (N - 1) // N
        if N == 1:
            for rc, sd in zip(recvs, sends):
                rc.copy_(sd)
            return [_Done()]
        if self.mode == "a2a":
            return [dist.all_to_all_single(rc, sd, group=self.group, async_op=True) for rc, sd in zip(recvs, sends)]
        ops = []
        for rc, sd in zip(recvs, sends):
            rc[r].copy_(sd[r])                                   # my own chunk never touches the fabric
            for step in range(1, N):                             # same peer order on every rank, sends and receives
                to, frm = (r + step) % N, (r - step) % N         # of one step pair up: no rank waits on a busy peer
                ops.append(dist.P2POp(dist.isend, sd[to], dist.get_global_rank(self.group, to), self.group))
                ops.append(dist.P2POp(dist.irecv, rc[frm], dist.get_global_rank(self.group, frm), self.group))
        return dist.batch_isend_irecv(ops)

    def all_gather(self, out, x):
        """out [N, ...] <- x from every rank."""
        self.calls += 1
        self.bytes_out += x.numel() * x.element_size() * (self.size() - 1)
        if self.size() == 1:
            out[0].copy_(x)
            return _Done()
        x = x.contiguous()   # (gloo wants the output as the inputs stacked along dim 0 of the INPUT's rank)
        return dist.all_gather_into_tensor(out.view((-1,) + tuple(x.shape[1:])), x, group=self.group, async_op=True)


class NoFabricExchange:
    """Measurement aid (bench.py `roofline_xgmi.exposed_ms`): the exchanges of one rank of an N-rank job with every
    transfer replaced by a local copy of the same shape -- same kernels, same launch structure, no fabric.  The RESULTS are
    wrong by construction (every peer's chunk is a copy of this rank's); only the clock is read: step time with the real
    exchange minus step time with this one = the exchange time the step could not hide."""

    def __init__(self, n, rank=0):
        self.n, self.r = n, rank
        self.bytes_out = 0
        self.calls = 0

    def size(self):
        return self.n

    def rank(self):
        return self.r

    def all_to_all(self, recvs, sends):
        for rc, sd in zip(recvs, sends):
            rc.copy_(sd)
        return [_Done()]

    def all_gather(self, out, x):
        out.copy_(x.unsqueeze(0).expand_as(out))
        return _Done()


def _wait_all(works):
    for w in works:
        w.wait()


def _hip_prologue(xq, xk, xv, wq, wk, cos, sin, outs, Hn, head0, n_heads, s_rope):
    """RMSNorm + RoPE of Q, K and the head scatter of Q, K, V in one HIP launch (jenga_sp_qkv_prologue); (xq, xk) or xv
    may be None together with their outputs."""
    _capi.sp_qkv_prologue(xq, xk, xv, wq, wk, cos, sin, outs[0], outs[1], outs[2], Hn, head0=head0, n_heads=n_heads,
                          s_rope=s_rope)


class UlyssesAttenCarve(torch.nn.Module):
    """Callable with the signature of xFuserLongContextAttention.forward (xdit_ring_atten.py:61-85); assign an
    instance to `block.hybrid_seq_parallel_attn` exactly as jenga_hyvideo_multigpu.py:181-182 does.

    Entry points:
      forward(...)      the reference's signature: already normalised / rotated q, k, v of the local shard
      forward_qkv(...)  the RAW q / k / v slices of the QKV GEMM outputs: per-head RMSNorm, RoPE and the peer-major
                        pack of all three happen in ONE kernel per stream (image | text), the text call writing this
                        rank's head slice straight into the attention inputs
      begin(...)        (jenga_amd.dit's blocks) -> PendingAttenCarve: the same steps, posted one by one so that the
                        caller's GEMMs run while the exchanges are in flight

    The local steps are injectable so that the world_size > 1 exchange logic can be exercised on CPU tensors over gloo
    against the oracle (tests/test_ulysses_gloo.py supplies oracle stand-ins); the defaults are the HIP kernels and
    raise on CPU tensors -- there is no CPU path in the product:
      select_fn(q_all, k_all, top_k, text_blocks, p, neighbors) -> (idx, cnt)
      attend_fn(q_all, k_all, v_all, idx, cnt, seqlens, text_blocks, text_amp) -> [1,S,H/N,D]
      pack_fn(t [B,S_loc,H,D], N) -> [N,B,S_loc,H/N,D];  unpack_fn(recv [N,B,S_loc,H/N,D], N, out [B,S_loc,H,D])
      prologue_fn(xq, xk, xv, wq, wk, cos, sin, (oq, ok, ov), H/N, head0, n_heads, s_rope)"""

    def __init__(self, group=None, select_fn=None, attend_fn=None, pack_fn=None, unpack_fn=None, exchange=None,
                 prologue_fn=None, pipeline=None):
        super().__init__()
        self.group = group
        # one head at a time (module docstring, round 5); default from JENGA_ULYSSES_PIPELINE
        self.pipeline = (os.environ.get("JENGA_ULYSSES_PIPELINE", "0") == "1") if pipeline is None else bool(pipeline)
        self.select_fn = select_fn or _hip_select
        self.attend_fn = attend_fn or _hip_attend
        self.pack_fn = pack_fn or _pack_heads
        self.unpack_fn = unpack_fn or _unpack_heads
        self.prologue_fn = prologue_fn or _hip_prologue
        self._exchange = exchange

    def exchange(self):
        if self._exchange is None:
            self._exchange = DistExchange(self.group if self.group is not None else get_sp_group().group)
        return self._exchange

    # ---- local stages ------------------------------------------------------------------------------------------
    def stage_out(self, o_recv, txt_all, N, S_loc, S_txt, dtype, device, out=None):
        """o_recv [N,S_loc,Hn,D] (chunk p = my tokens, rank p's heads), txt_all [N,1,S_txt,Hn,D] -> [1,S_loc+S_txt,H,D]
        (written into `out` when given: may be a strided view, e.g. the left part of linear2's concat buffer).
        Pipelined call: lists of per-head tensors (Hn entries with one head each); head g of every peer p is head
        p * Hn + g of the result -- a strided head view of it."""
        if isinstance(o_recv, (list, tuple)):
            G, D = len(o_recv), o_recv[0].shape[-1]
            result = out if out is not None else torch.empty((1, S_loc + S_txt, N * G, D), dtype=dtype, device=device)
            for g in range(G):
                self.unpack_fn(o_recv[g].view(N, 1, S_loc, 1, D), N, result[:, :S_loc, g::G])
                self.unpack_fn(txt_all[g], N, result[:, S_loc:, g::G])
            return result
        Hn, D = o_recv.shape[-2:]
        result = out if out is not None else torch.empty((1, S_loc + S_txt, N * Hn, D), dtype=dtype, device=device)
        self.unpack_fn(o_recv.view(N, 1, S_loc, Hn, D), N, result[:, :S_loc])
        self.unpack_fn(txt_all, N, result[:, S_loc:])
        return result

    def _check(self, B, S_loc, H, S_txt, N):
        if B != 1:
            raise ValueError("jenga_amd Ulysses: batch must be 1")
        if H % N:
            raise ValueError(f"heads ({H}) must be divisible by the sequence-parallel degree ({N})")
        if (S_loc * N) % 128 or S_txt % 128:
            raise ValueError("gathered image length and text length must be multiples of 128")

    def begin(self, B, S_loc, H, S_txt, dtype, device, D=128, pipeline=None):
        """Start one attention call of the sequence-parallel blocks: allocates the peer-major send buffers and the
        gathered attention inputs and returns the pending call (`PendingAttenCarve`).  The caller posts Q, K (and V)
        as soon as their GEMM is done, keeps issuing independent work (the V GEMM, the text stream, the MLP half of
        linear1) while the exchange is in flight on RCCL's stream, and calls .finish()."""
        ex = self.exchange()
        N, r = ex.size(), ex.rank()
        self._check(B, S_loc, H, S_txt, N)
        return PendingAttenCarve(self, ex, N, r, B, S_loc, H, S_txt, D, dtype, device,
                                 pipeline=self.pipeline if pipeline is None else bool(pipeline))

    @torch.no_grad()
    def forward(self, attn, query, key, value, *, joint_tensor_query=None, joint_tensor_key=None,
                joint_tensor_value=None, dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1),
                alibi_slopes=None, deterministic=False, return_attn_probs=False, joint_strategy="none", top_k=0,
                text_amp=0.0, block_neighbor_list=None, p_remain_rates=0.0, cu_seqlens_q=None, cu_seqlens_kv=None,
                dense=False):
        if joint_strategy != "rear" or joint_tensor_query is None or joint_tensor_key is None \
                or joint_tensor_value is None:
            raise ValueError("jenga_amd Ulysses: only joint_strategy='rear' with text q/k/v (the Jenga call) is supported")
        B, S_loc, H, D = query.shape
        S_txt = joint_tensor_query.shape[1]
        # (the reference-signature call has no head-group pipeline: one group whatever JENGA_ULYSSES_PIPELINE says -- ADVICE r5)
        pend = self.begin(B, S_loc, H, S_txt, query.dtype, query.device, D, pipeline=False)
        pend.post_packed(query, key, value, joint_tensor_query, joint_tensor_key, joint_tensor_value)
        return pend.finish(top_k=top_k, text_amp=text_amp, block_neighbor_list=block_neighbor_list,
                           p_remain_rates=p_remain_rates, cu_seqlens_q=cu_seqlens_q, dense=dense)

    @torch.no_grad()
    def forward_qkv(self, img_qkv, txt_qkv, img_norm_w, txt_norm_w, freqs_cis, *, top_k=0, text_amp=0.0,
                    block_neighbor_list=None, p_remain_rates=0.0, cu_seqlens_q=None, out=None):
        """The fused entry in one call: img_qkv / txt_qkv = (q, k, v) RAW slices of the QKV GEMM outputs of the local
        image shard [1,S_loc,H,D] and of the (replicated) text rows [1,S_txt,H,D]; *_norm_w = (q_norm.weight,
        k_norm.weight); freqs_cis = (cos, sin) rows of the local shard.  Same result as forward() on the
        normalised / rotated tensors, bit for bit; returns [1, S_loc + S_txt, H, D] (in `out` when given).
        (jenga_amd.dit's blocks use begin() / post_*() / finish() themselves to put GEMMs between the steps.)"""
        B, S_loc, H, D = img_qkv[0].shape
        pend = self.begin(B, S_loc, H, txt_qkv[0].shape[1], img_qkv[0].dtype, img_qkv[0].device, D)
        pend.post_qkv(img_qkv[0], img_qkv[1], img_qkv[2], img_norm_w, freqs_cis)
        pend.put_text(txt_qkv[0], txt_qkv[1], txt_qkv[2], txt_norm_w)
        return pend.finish(top_k=top_k, text_amp=text_amp, block_neighbor_list=block_neighbor_list,
                           p_remain_rates=p_remain_rates, cu_seqlens_q=cu_seqlens_q, out=out)


class PendingAttenCarve:
    """One sequence-parallel attention call between its first posted exchange and its result (exchange / compute
    overlap; the reference runs everything on one stream, xdit_ring_atten.py:118-131, 212-217).

      post_qk(q, k, (wq, wk), (cos, sin))   prologue kernel (RMSNorm + RoPE + peer-major pack) + Q, K exchange
      post_v(v)                             pack + V exchange
      post_qkv(q, k, v, ...)                both from ONE prologue launch (one QKV GEMM in front)
      put_text(q, k, v, (wq, wk))           local: this rank's head slice of the replicated text rows, in place
      finish(..., out=, while_out=)         wait Q, K -> pool + select -> wait V -> re-tile + attention -> post the
                                            O exchange -> while_out() (caller's independent work) -> wait -> unpack

    Everything between a post_* and finish() that the caller enqueues on the compute stream overlaps the transfer."""

    def __init__(self, sp, ex, N, r, B, S_loc, H, S_txt, D, dtype, device, pipeline=None):
        self.sp, self.ex, self.N, self.r = sp, ex, N, r
        pipeline = sp.pipeline if pipeline is None else pipeline
        self.B, self.S_loc, self.H, self.S_txt, self.D = B, S_loc, H, S_txt, D
        self.Hn, self.S_img = H // N, S_loc * N
        self.dtype, self.device = dtype, device
        # head groups: 1 (all H/N heads of the rank in one exchange / one attention launch) or, pipelined, H/N groups of one
        # head.  Everything is allocated group-major -- [G, B, S, hg, D] attention inputs, [G, N, B, S_loc, hg, D] send
        # buffers -- so that a group's chunks are contiguous messages; with G == 1 that IS the round-4 layout.
        self.G = self.Hn if (pipeline and self.Hn > 1) else 1
        self.hg = self.Hn // self.G
        mk = lambda shape: torch.empty(shape, dtype=dtype, device=device)
        self.alloc = [mk((self.G, B, self.S_img + S_txt, self.hg, D)) for _ in range(3)]
        self.fulls = [[a[g] for a in self.alloc] for g in range(self.G)]                  # [g][q|k|v] -> [B, S, hg, D]
        self.recvs = [[f[0, :self.S_img].view(N, S_loc, self.hg, D) for f in fg] for fg in self.fulls]
        self.sends = [None, None, None]
        self.w_qk = self.w_v = None

    def _send_buffers(self, which):
        for i in which:
            self.sends[i] = torch.empty((self.G, self.N, self.B, self.S_loc, self.hg, self.D), dtype=self.dtype,
                                        device=self.device)
        return [self._peer_major(self.sends[i]) for i in which]

    def _peer_major(self, t):
        """[G, N, B, S_loc, hg, D] -> the [N, B, S_loc, H/N, D] view the prologue kernel writes through its strides
        (contiguous for G == 1; for one head per group the head axis IS the group axis)."""
        return t[0] if self.G == 1 else t[:, :, :, :, 0].permute(1, 2, 3, 0, 4)

    def _text_view(self, i):
        """[B, S_txt, H/N, D] view of the text rows of attention input i over all groups."""
        a = self.alloc[i]
        return a[0][:, self.S_img:] if self.G == 1 else a[:, :, self.S_img:, 0].permute(1, 2, 0, 3)

    def _views(self, which, g):
        return [self.sends[i][g].view(self.N, self.S_loc, self.hg, self.D) for i in which]

    def _post(self, which):
        """One exchange per head group, in group order: group g's messages are complete before group g + 1's."""
        return [self.ex.all_to_all([self.recvs[g][i] for i in which], self._views(which, g)) for g in range(self.G)]

    def post_qk(self, q, k, norm_w, freqs_cis):
        cos, sin = freqs_cis
        oq, ok = self._send_buffers((0, 1))
        self.sp.prologue_fn(q, k, None, norm_w[0], norm_w[1], cos, sin, [oq, ok, None], self.Hn, 0, self.H, self.S_loc)
        self.w_qk = self._post((0, 1))

    def post_v(self, v):
        (ov,) = self._send_buffers((2,))
        self.sp.prologue_fn(None, None, v, None, None, None, None, [None, None, ov], self.Hn, 0, self.H, 0)
        self.w_v = self._post((2,))

    def post_qkv(self, q, k, v, norm_w, freqs_cis):
        cos, sin = freqs_cis
        outs = self._send_buffers((0, 1, 2))
        self.sp.prologue_fn(q, k, v, norm_w[0], norm_w[1], cos, sin, outs, self.Hn, 0, self.H, self.S_loc)
        # Q + K first, V behind them on the communication stream (the V transfer overlaps pooling + selection); pipelined:
        # group by group, so that head g is complete before head g + 1 starts to arrive
        if self.G == 1:
            self.w_qk = self._post((0, 1))
            self.w_v = self._post((2,))
        else:
            self.w_qk, self.w_v = [], []
            for g in range(self.G):
                self.w_qk.append(self.ex.all_to_all([self.recvs[g][0], self.recvs[g][1]], self._views((0, 1), g)))
                self.w_v.append(self.ex.all_to_all([self.recvs[g][2]], self._views((2,), g)))

    def put_text(self, q, k, v, norm_w):
        outs = [self._text_view(i) for i in range(3)]
        if v.stride() == q.stride():
            self.sp.prologue_fn(q, k, v, norm_w[0], norm_w[1], None, None, outs, self.Hn, self.r * self.Hn, self.Hn, 0)
        else:       # Q|K and V come from two GEMM outputs with different row strides: one launch each
            self.sp.prologue_fn(q, k, None, norm_w[0], norm_w[1], None, None, [outs[0], outs[1], None], self.Hn,
                                self.r * self.Hn, self.Hn, 0)
            self.sp.prologue_fn(None, None, v, None, None, None, None, [None, None, outs[2]], self.Hn,
                                self.r * self.Hn, self.Hn, 0)

    def post_packed(self, query, key, value, jq, jk, jv):
        """The reference-signature path: already normalised / rotated tensors, separate pack kernels."""
        if self.G != 1:
            raise RuntimeError("the reference-signature call (forward) has no head-group pipeline: use begin() / forward_qkv")
        hs = slice(self.r * self.Hn, (self.r + 1) * self.Hn)
        for i, (t, joint) in enumerate(((query, jq), (key, jk), (value, jv))):
            self.sends[i] = self.sp.pack_fn(t, self.N).view(1, self.N, self.B, self.S_loc, self.Hn, self.D)
            self.fulls[0][i][:, self.S_img:] = joint[:, :, hs]    # text is replicated on every rank: slice my heads
        self.w_qk = self._post((0, 1))
        self.w_v = self._post((2,))

    def finish(self, *, top_k=0, text_amp=0.0, block_neighbor_list=None, p_remain_rates=0.0, cu_seqlens_q=None,
               out=None, while_out=None, dense=False):
        if self.w_qk is None or self.w_v is None:
            raise RuntimeError("PendingAttenCarve.finish(): Q, K and V have not all been posted")
        sp, ex, N = self.sp, self.ex, self.N
        S_img, S_loc, S_txt, hg, D = self.S_img, self.S_loc, self.S_txt, self.hg, self.D
        dev, dt = self.device, self.dtype
        # cu_seqlens = [0, n_valid_text + S_img, S] (xdit_ring_atten.py:105,183-184) -- stays on the device
        seqlens = (cu_seqlens_q[1:2].to(torch.int64) - S_loc + S_img).to(device=dev, dtype=torch.int32)
        o_recvs, txt_alls, waits = [], [], []
        for g in range(self.G):
            q_all, k_all, v_all = self.fulls[g]
            _wait_all(self.w_qk[g])
            if dense:       # parallel_attention: no selection, every row masked at the valid length
                # (HIP only: the dense call goes straight to the library -- an injected select_fn / attend_fn, which exist for
                # the CPU tests of the AttenCarve path, is NOT consulted here; on CPU tensors this raises JengaError)
                _wait_all(self.w_v[g])
                o = _hip_attend_dense(q_all, k_all, v_all, seqlens)
            else:
                idx, cnt = sp.select_fn(q_all, k_all, top_k, S_txt // 128, p_remain_rates, block_neighbor_list)
                _wait_all(self.w_v[g])                             # the V transfer overlapped pooling + selection
                o = sp.attend_fn(q_all, k_all, v_all, idx, cnt, seqlens, S_txt // 128, text_amp)
            # ---- exchange out: image rows (already peer-major: chunk p = rank p's tokens) back to sequence shards;
            #      text rows gathered over heads (the reference repeats them N times and all-to-alls, :206-217).
            #      Pipelined: head g's O exchange is in flight while head g + 1 is attended
            o_img = o[0, :S_img].reshape(N, S_loc, hg, D)
            if not o_img.is_contiguous():
                o_img = o_img.contiguous()
            o_recv = torch.empty((N, S_loc, hg, D), dtype=dt, device=dev)
            txt_all = torch.empty((N, self.B, S_txt, hg, D), dtype=dt, device=dev)
            waits.append((ex.all_to_all([o_recv], [o_img]), ex.all_gather(txt_all, o[:, S_img:])))
            o_recvs.append(o_recv)
            txt_alls.append(txt_all)
        if while_out is not None:
            while_out()                                        # the caller's work that needs neither O nor its buffers
        for w_o, w_t in waits:
            _wait_all(w_o)
            w_t.wait()
        self.sends = self.fulls = self.recvs = self.alloc = None
        if self.G == 1:
            return sp.stage_out(o_recvs[0], txt_alls[0], N, S_loc, S_txt, dt, dev, out=out)
        return sp.stage_out(o_recvs, txt_alls, N, S_loc, S_txt, dt, dev, out=out)


# name the reference uses (jenga_hyvideo_multigpu.py:181)
xFuserLongContextAttention = UlyssesAttenCarve
