"""ctypes binding of libjenga_amd.so (include/jenga_amd.h).  No compute happens in Python: every function
enqueues HIP kernels on torch's current stream and raises if the library is missing or a call fails.
There is NO CPU fallback on purpose: the product path must fail loudly when the HIP extension is absent."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("JENGA_LIB", os.path.join(_HERE, "libjenga_amd.so"))

JENGA_BF16, JENGA_FP16 = 0, 1
# jenga_bsattn_fwd flags (include/jenga_amd.h).  No kernel bit = the round-1 kernel, exactly as in the C header.
ATTN_XCD_REMAP = 1
ATTN_BALANCE = 4     # query blocks (pairs) DRAWN from per-XCD queues on an oversubscribed grid: evens out the speed differences
#                      between the XCDs of a chip; bit-identical; in the default since round 4
ATTN_LEGACY = 0      # (readability alias) no kernel bit: the round-1 kernel, one query block per 4-wave workgroup
ATTN_LP = 8          # round-1 decomposition + in-wave software pipeline (csrc/bsattn3.hip)
ATTN_SORTED = 16     # (Python-side) kept-count-aware launch order: jenga_order_by_count feeds jenga_bsattn_fwd's `order`
ATTN_PAIR = 64       # (Python-side) route to jenga_pair_merge + jenga_bsattn_pair_fwd: the pair kernel of round 5
#                      (csrc/bsattn5.hip: two Hilbert-adjacent query blocks per workgroup, 64 query rows per wave)
# LP default: LP kernel, XCD remap, kept-count-aware order inside every XCD's range (+3.3 % sustained on lists whose counts
# vary: profiles/r03_attn_order_ab.json), cross-XCD balancing (-1.7 % loop time: profiles/r04_attn_balance_ab.json).
ATTN_LP_FLAGS = ATTN_XCD_REMAP | ATTN_BALANCE | ATTN_LP | ATTN_SORTED
ATTN_PAIR_FLAGS = ATTN_XCD_REMAP | ATTN_BALANCE | ATTN_SORTED | ATTN_PAIR
ATTN_DEFAULT_FLAGS = int(os.environ.get("JENGA_ATTN_FLAGS", str(ATTN_LP_FLAGS)))


SELECT_DEVICE_SCAN = 1   # jenga_block_select flags: torch's DEVICE cumsum semantics for the kept-count rule
SELECT_DEFAULT_FLAGS = int(os.environ.get("JENGA_SELECT_FLAGS", "0"))

_vp, _i64, _i32, _f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float

# name -> (restype, argtypes): every symbol the header declares
SIGNATURES = {
    "jenga_abi_version": (_i32, []),
    "jenga_last_error": (ctypes.c_char_p, []),
    "jenga_gilbert_map": (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "jenga_gilbert_neighbors": (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "jenga_gather_rows": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64]),
    "jenga_rmsnorm_rope": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp] + [_i64] * 10 + [_f32, _i32]),
    "jenga_rmsnorm_rows": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _f32, _i32, _i32]),
    "jenga_rope_complex": (_i32, [_vp, _vp, _vp, _vp, _vp] + [_i64] * 10 + [_i32, _i32]),
    "jenga_wan_norm_rope": (_i32, [_vp] * 6 + [_i64] * 5 + [_f32, _i32]),
    "jenga_ln_modulate": (_i32, [_vp] * 8 + [_i64] * 4 + [_f32, _i32]),
    "jenga_gate_residual": (_i32, [_vp] * 7 + [_i64] * 5 + [_i32]),
    "jenga_gelu_tanh": (_i32, [_vp, _vp, _vp] + [_i64] * 4 + [_i32]),
    "jenga_wan_ln_modulate": (_i32, [_vp] * 7 + [_i64] * 4 + [_f32, _i32, _i32]),
    "jenga_wan_gate_residual": (_i32, [_vp] * 5 + [_i64] * 5 + [_i32]),
    "jenga_qk_norm_rope_pool": (_i32, [_vp] * 11 + [_i64] * 13 + [_f32, _i32]),
    "jenga_sp_qkv_prologue": (_i32, [_vp] * 11 + [_i64] * 14 + [_f32, _i32]),
    "jenga_linear": (_i32, [_vp] * 7 + [_i64] * 7 + [_i32, _vp, _i64, _i32]),
    "jenga_block_pool": (_i32, [_vp, _vp, _vp] + [_i64] * 6 + [_i32]),
    "jenga_block_select": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp] + [_i64] * 6 + [_f32, _i64, _i32, _i32]),
    "jenga_order_by_count": (_i32, [_vp, _vp, _i64, _i64, _i64, _vp]),
    "jenga_pack_v_bytes": (ctypes.c_size_t, [_i64, _i64, _i64]),
    "jenga_pack_v": (_i32, [_vp, _vp, _vp] + [_i64] * 8 + [_i32]),
    "jenga_bsattn_fwd": (_i32, [_vp] * 9 + [_i64] * 13 + [_f32, _f32, _i64, _i32, _i32]),
    "jenga_ulysses_pack_heads": (_i32, [_vp, _vp, _vp] + [_i64] * 7),
    "jenga_ulysses_unpack_heads": (_i32, [_vp, _vp, _vp] + [_i64] * 7),
    "jenga_stream_delay": (_i32, [_vp, ctypes.c_double]),
    "jenga_cross_attn_fwd": (_i32, [_vp] * 5 + [_i64] * 14 + [_f32, _i32]),
    "jenga_linear_export_choices": (_i64, [_vp, _i64]),
    "jenga_linear_import_choices": (_i32, [_vp, _i64]),
    "jenga_linear_import_mismatches": (_i64, []),
    "jenga_pair_merge": (_i32, [_vp, _vp, _vp] + [_i64] * 4 + [_vp, _vp]),
    "jenga_bsattn_pair_fwd": (_i32, [_vp] * 9 + [_i64] * 13 + [_f32, _f32, _i64, _i32, _i32]),
}

_lib = None


class AttnProfile:
    """Optional per-launch timing of jenga_bsattn_fwd with HIP events recorded on the launch stream (bench.py's
    roofline leg).  Enable with `_capi.ATTN_PROFILE = AttnProfile()`; read with .summary() after a synchronize."""

    def __init__(self):
        self.events = []        # (start, stop)
        self.pairs = None       # device int64 scalar: kept (q-block, kv-block) pairs over all recorded launches
        self.launches = 0
        self.last_lists = None  # (idx, cnt) of the most recent launch with image query blocks
        self.tag = None         # set by the caller (bench.py: the step's sa-drop rate); recorded per launch
        self.per_launch = []    # (tag, device scalar: pairs of the launch), aligned with `events`

    def summary(self):
        ms = sum(a.elapsed_time(b) for a, b in self.events)
        out = dict(launches=self.launches, total_ms=ms, pairs=int(self.pairs.item()) if self.pairs is not None else 0)
        by_tag = {}
        for (tag, pr), (a, b) in zip(self.per_launch, self.events):
            t = by_tag.setdefault(tag, dict(launches=0, total_ms=0.0, pairs=0))
            t["launches"] += 1
            t["total_ms"] += a.elapsed_time(b)
            t["pairs"] += int(pr.item()) if torch.is_tensor(pr) else int(pr)
        out["by_tag"] = by_tag
        if self.last_lists is not None:
            # share of a query block's kept image kv blocks that the NEXT query block of the same head keeps too
            # (how coherent the lists are: ~0.35 for random lists at 30 % density, ~0.9 for a trained model's)
            idx, cnt = self.last_lists
            B, H, nq, nb = idx.shape
            valid = torch.arange(nb, device=idx.device)[None, None, None, :] < cnt[..., None]
            hit = torch.zeros((B, H, nq, nb), dtype=torch.int8, device=idx.device)
            hit.scatter_add_(-1, torch.where(valid, idx, torch.zeros_like(idx)).long(), valid.to(torch.int8))
            img = hit[..., :nq].bool()
            both = (img[:, :, :-1] & img[:, :, 1:]).sum(-1).float()
            out["adjacent_shared_frac"] = float((both / img[:, :, :-1].sum(-1).clamp(min=1).float()).mean().item())
        return out


ATTN_PROFILE = None


class JengaError(RuntimeError):
    pass


def lib():
    """Load the shared library (once).  Raises JengaError with build instructions if it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise JengaError(
                f"{LIB_PATH} is missing: build it with `python -m jenga_amd.build` (hipcc --offload-arch=gfx950). "
                "jenga_amd has no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        if L.jenga_abi_version() != 4:
            raise JengaError("libjenga_amd.so ABI version mismatch; rebuild")
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        raise JengaError(f"{what} failed (code {rc}): {lib().jenga_last_error().decode()}")


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


class _Here:
    """No-op context: the tensor's device is already the current one (the common case: one process per GPU)."""

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_HERE_CTX = _Here()
_ALWAYS_GUARD = os.environ.get("JENGA_DEVICE_GUARD", "") == "always"     # (A/B switch: round 3's behaviour)


def _on(device):
    """`with torch.cuda.device(device)` only when it would change anything: the guard's two set_device calls cost more
    host time than the ctypes call they surround (60 layers x ~30 launches per step; the per-rank steps of the
    low-resolution stages of an 8-rank job are host-bound)."""
    if not _ALWAYS_GUARD and (device.index is None or device.index == torch.cuda.current_device()):
        return _HERE_CTX
    return torch.cuda.device(device)


def _need_gpu(t, what):
    if not t.is_cuda:
        raise JengaError(f"{what}: tensors must live on the GPU (got {t.device}); jenga_amd has no CPU path")


def dtype_code(dt):
    if dt == torch.bfloat16:
        return JENGA_BF16
    if dt == torch.float16:
        return JENGA_FP16
    raise ValueError(f"jenga_amd supports bfloat16 and float16 only, got {dt}")


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _bshd_strides(t):
    """[B,S,H,D] tensor with unit innermost stride -> (sb, ss, sh) in elements."""
    if t.dim() != 4 or t.stride(3) != 1:
        raise ValueError("expected a [B,S,H,D] tensor with contiguous head_dim")
    return t.stride(0), t.stride(1), t.stride(2)


# ------------------------------------------------------------------------------------------------ geometry
def gilbert_map(t, h, w, sliced, device):
    n = t * h * w
    l2h = torch.empty(n, dtype=torch.int64, device=device)
    h2l = torch.empty(n, dtype=torch.int64, device=device)
    _need_gpu(l2h, "gilbert_map")
    with _on(l2h.device):
        _check(lib().jenga_gilbert_map(_stream(l2h.device), t, h, w, int(bool(sliced)), _p(l2h), _p(h2l)),
               "jenga_gilbert_map")
    return l2h, h2l


def gilbert_neighbors(t, h, w, block, l2h):
    _need_gpu(l2h, "gilbert_neighbors")
    n = t * h * w
    nb = (n + block - 1) // block
    out = torch.zeros((nb, nb), dtype=torch.uint8, device=l2h.device)
    with _on(l2h.device):
        _check(lib().jenga_gilbert_neighbors(_stream(l2h.device), t, h, w, block, _p(l2h), _p(out)),
               "jenga_gilbert_neighbors")
    return out.view(torch.bool)


# ------------------------------------------------------------------------------------------------ row ops
def gather_rows(src, index, out=None):
    """src [B, N, C] contiguous rows; index int64 [M] on the same device -> out [B, M, C] = src[:, index]."""
    _need_gpu(src, "gather_rows")
    if src.dim() != 3 or not src.is_contiguous():
        raise ValueError("gather_rows expects a contiguous [B, N, C] tensor")
    if index.dtype != torch.int64 or index.device != src.device:
        raise ValueError("index must be an int64 tensor on the same device")
    B, N, C = src.shape
    M = index.numel()
    if out is None:
        out = torch.empty((B, M, C), dtype=src.dtype, device=src.device)
    rb = C * src.element_size()
    with _on(src.device):
        _check(lib().jenga_gather_rows(_stream(src.device), _p(src), _p(out), _p(index), B, M, rb, N * rb, M * rb),
               "jenga_gather_rows")
    return out


def rmsnorm_rope(x, weight, cos, sin, s_rope=None, eps=1e-6, out=None):
    """x [B,S,H,128] (any B/S/H strides); weight [128] or None; cos/sin fp32 [>=s_rope,128] or None."""
    _need_gpu(x, "rmsnorm_rope")
    if x.shape[-1] != 128:
        raise ValueError("head_dim must be 128")
    B, S, H, _ = x.shape
    if out is None:
        out = torch.empty((B, S, H, 128), dtype=x.dtype, device=x.device)
    xs, os_ = _bshd_strides(x), _bshd_strides(out)
    if weight is not None:
        weight = weight.to(device=x.device, dtype=x.dtype).contiguous()
    if cos is not None:
        if cos.dtype != torch.float32 or sin.dtype != torch.float32 or cos.shape[-1] != 128:
            raise ValueError("cos/sin must be float32 [S,128]")
        cos, sin = cos.contiguous(), sin.contiguous()
        if s_rope is None:
            s_rope = cos.shape[0]
        if s_rope > cos.shape[0] or s_rope > S:
            raise ValueError("s_rope exceeds the table / sequence length")
    else:
        s_rope = 0
    with _on(x.device):
        _check(lib().jenga_rmsnorm_rope(_stream(x.device), _p(x), _p(out), _p(weight), _p(cos), _p(sin), B, S, H,
                                        *xs, *os_, s_rope, float(eps), dtype_code(x.dtype)), "jenga_rmsnorm_rope")
    return out


def qk_norm_rope_pool(xq, xk, wq, wk, cos, sin, out_q, out_k, s_rope=None, qpool=None, kpool=None, pool_block0=0,
                      eps=1e-6):
    """Fused per-head RMSNorm + RoPE of Q and K (+ 128-token block means of the results).
    xq, xk [B,S,H,128] with the SAME strides (the q / k slices of one QKV GEMM output), S a multiple of 128;
    out_q, out_k [B,S,H,128] with the same strides as each other; qpool [B,H,nq,128] / kpool [B,H,nk,128] receive the
    means of blocks pool_block0 .. pool_block0 + S/128 - 1 (where inside the pooled tensor)."""
    _need_gpu(xq, "qk_norm_rope_pool")
    B, S, H, D = xq.shape
    if D != 128 or S % 128 or xk.shape != xq.shape or out_q.shape != xq.shape or out_k.shape != xq.shape:
        raise ValueError("qk_norm_rope_pool: q / k / outputs must be [B, n*128, H, 128] of one shape")
    if _bshd_strides(xq) != _bshd_strides(xk) or _bshd_strides(out_q) != _bshd_strides(out_k):
        raise ValueError("qk_norm_rope_pool: q and k (and their outputs) must share their strides")
    if H > 64:
        raise ValueError("qk_norm_rope_pool: at most 64 heads")
    wq = None if wq is None else wq.to(device=xq.device, dtype=xq.dtype).contiguous()
    wk = None if wk is None else wk.to(device=xq.device, dtype=xq.dtype).contiguous()
    if cos is not None:
        if cos.dtype != torch.float32 or sin.dtype != torch.float32 or cos.shape[-1] != 128:
            raise ValueError("cos/sin must be float32 [S,128]")
        cos, sin = cos.contiguous(), sin.contiguous()
        if s_rope is None:
            s_rope = cos.shape[0]
        if s_rope > cos.shape[0] or s_rope > S:
            raise ValueError("s_rope exceeds the table / sequence length")
    else:
        s_rope = 0
    nq = nk = 0
    if qpool is not None:
        if qpool.dim() != 4 or qpool.shape[0] != B or qpool.shape[1] != H or qpool.shape[3] != 128 \
                or not qpool.is_contiguous() or qpool.dtype != xq.dtype:
            raise ValueError("qpool must be a contiguous [B,H,nq,128] tensor of the input dtype")
        nq = qpool.shape[2]
    if kpool is not None:
        if kpool.dim() != 4 or kpool.shape[0] != B or kpool.shape[1] != H or kpool.shape[3] != 128 \
                or not kpool.is_contiguous() or kpool.dtype != xq.dtype:
            raise ValueError("kpool must be a contiguous [B,H,nk,128] tensor of the input dtype")
        nk = kpool.shape[2]
    with _on(xq.device):
        _check(lib().jenga_qk_norm_rope_pool(_stream(xq.device), _p(xq), _p(xk), _p(out_q), _p(out_k), _p(wq), _p(wk),
                                             _p(cos), _p(sin), _p(qpool), _p(kpool), B, S // 128, H,
                                             *_bshd_strides(xq), *_bshd_strides(out_q), int(s_rope), int(pool_block0),
                                             nq, nk, float(eps), dtype_code(xq.dtype)), "jenga_qk_norm_rope_pool")
    return out_q, out_k


def sp_qkv_prologue(xq, xk, xv, wq, wk, cos, sin, out_q, out_k, out_v, heads_per_peer, head0=0, n_heads=None,
                    s_rope=None, eps=1e-6):
    """Sequence-parallel prologue: per-head RMSNorm + RoPE of Q and K and the Ulysses head scatter of Q, K, V in one
    launch.  xq, xk, xv [B,S,H,128] with the SAME strides (any S).  Outputs, all of one shape and stride set:
      peer-major send buffers [N,B,S,H/N,128] (head0 = 0, all heads; contiguous or any 16-byte-aligned strides), or
      [B,S,n_heads,128] views (any strides) receiving the head window [head0, head0 + n_heads) -- a rank's own slice
      of the replicated text rows, written in place behind the gathered image rows.
    Either (xq, xk, out_q, out_k) or (xv, out_v) may be None: the blocks run the Q|K and the V GEMM separately and post
    the Q, K exchange before the V GEMM (exchange / compute overlap)."""
    has_qk, has_v = xq is not None, xv is not None
    if not (has_qk or has_v) or (has_qk and (xk is None or out_q is None or out_k is None)) \
            or (not has_qk and xk is not None) or (has_v and out_v is None):
        raise ValueError("sp_qkv_prologue: pass (xq, xk, out_q, out_k) and / or (xv, out_v)")
    xs = [t for t in (xq, xk, xv) if t is not None]
    outs = ([out_q, out_k] if has_qk else []) + ([out_v] if has_v else [])
    x0, o0 = xs[0], outs[0]
    _need_gpu(x0, "sp_qkv_prologue")
    B, S, H, D = x0.shape
    if D != 128 or any(t.shape != x0.shape for t in xs):
        raise ValueError("sp_qkv_prologue: q / k / v must be [B, S, H, 128] of one shape")
    if any(_bshd_strides(t) != _bshd_strides(x0) for t in xs):
        raise ValueError("sp_qkv_prologue: q, k and v must share their strides")
    if n_heads is None:
        n_heads = H - head0
    Hn = int(heads_per_peer)
    if o0.dim() == 5:
        N = o0.shape[0]
        if head0 != 0 or n_heads != H or tuple(o0.shape) != (N, B, S, Hn, 128) or N * Hn != H \
                or o0.stride(4) != 1 or any(st % 8 for st in o0.stride()[:4]):
            raise ValueError("sp_qkv_prologue: peer-major outputs must be [N, B, S, H/N, 128] (unit inner stride; the "
                             "other strides multiples of 8 elements: contiguous, or the head-group-major view of the "
                             "pipelined sequence-parallel call)")
        o_sp, o_sb, o_ss, o_sh = o0.stride(0), o0.stride(1), o0.stride(2), o0.stride(3)
    else:
        if tuple(o0.shape) != (B, S, n_heads, 128) or n_heads > Hn or head0 % Hn + n_heads > Hn:
            raise ValueError("sp_qkv_prologue: head-window outputs must be [B, S, n_heads, 128] inside one peer's heads")
        o_sp = 0
        o_sb, o_ss, o_sh = _bshd_strides(o0)
        if head0 % Hn:      # the kernel writes head h at local index h % Hn: shift the base so that head0 lands on 0
            raise ValueError("sp_qkv_prologue: head0 must be a multiple of heads_per_peer")
    for t in outs:
        if t.shape != o0.shape or t.stride() != o0.stride() or t.dtype != x0.dtype:
            raise ValueError("sp_qkv_prologue: the outputs must share shape, strides and dtype")
    wq = None if wq is None else wq.to(device=x0.device, dtype=x0.dtype).contiguous()
    wk = None if wk is None else wk.to(device=x0.device, dtype=x0.dtype).contiguous()
    if cos is not None:
        if cos.dtype != torch.float32 or sin.dtype != torch.float32 or cos.shape[-1] != 128:
            raise ValueError("cos/sin must be float32 [S,128]")
        cos, sin = cos.contiguous(), sin.contiguous()
        if s_rope is None:
            s_rope = cos.shape[0]
        if s_rope > cos.shape[0] or s_rope > S:
            raise ValueError("s_rope exceeds the table / sequence length")
    else:
        s_rope = 0
    with _on(x0.device):
        _check(lib().jenga_sp_qkv_prologue(_stream(x0.device), _p(xq), _p(xk), _p(xv), _p(out_q), _p(out_k), _p(out_v),
                                           _p(wq), _p(wk), _p(cos), _p(sin), B, S, H, int(head0), int(n_heads), Hn,
                                           *_bshd_strides(x0), int(o_sp), int(o_sb), int(o_ss), int(o_sh),
                                           int(s_rope), float(eps), dtype_code(x0.dtype)), "jenga_sp_qkv_prologue")
    return out_q, out_k, out_v


def stream_delay(microseconds, stream=None, device=None):
    """Measurement aid (bench.py --simulate-ranks): keep `stream` (default: the current one) busy for that long."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    st = torch.cuda.current_stream(dev) if stream is None else stream
    with _on(dev):
        _check(lib().jenga_stream_delay(ctypes.c_void_p(st.cuda_stream), float(microseconds)), "jenga_stream_delay")


def rmsnorm_rows(x, weight, eps):
    """WanRMSNorm: x [..., C] (bf16/fp16, last dim contiguous), weight [C] fp32 or x.dtype -> x.dtype*weight.dtype."""
    _need_gpu(x, "rmsnorm_rows")
    C = x.shape[-1]
    x2 = x.reshape(-1, C)
    if x2.stride(-1) != 1:
        x2 = x2.contiguous()
    w32 = weight.dtype == torch.float32
    w = weight.to(device=x.device).contiguous() if w32 else weight.to(device=x.device, dtype=x.dtype).contiguous()
    out = torch.empty(x2.shape, dtype=torch.float32 if w32 else x.dtype, device=x.device)
    with _on(x.device):
        _check(lib().jenga_rmsnorm_rows(_stream(x.device), _p(x2), _p(out), _p(w), x2.shape[0], C, x2.stride(0),
                                        out.stride(0), float(eps), dtype_code(x.dtype), int(w32)),
               "jenga_rmsnorm_rows")
    return out.reshape(x.shape)


def rope_complex(x, cos64, sin64, s_rope, out_dtype=torch.float32):
    """x [B,S,H,128] (bf16/fp16/fp32); cos64/sin64 float64 [>=s_rope, 64] -> fp32 (rope_apply's result) or bf16."""
    _need_gpu(x, "rope_complex")
    B, S, H, D = x.shape
    if D != 128 or cos64.dtype != torch.float64 or cos64.shape[-1] != 64 or cos64.shape[0] < s_rope:
        raise ValueError("rope_complex: head_dim 128 and float64 [S,64] tables required")
    codes = {torch.bfloat16: 0, torch.float16: 1, torch.float32: 2}
    if x.dtype not in codes or out_dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("rope_complex: unsupported dtype")
    if x.stride(-1) != 1:
        x = x.contiguous()
    out = torch.empty((B, S, H, D), dtype=out_dtype, device=x.device)
    with _on(x.device):
        _check(lib().jenga_rope_complex(_stream(x.device), _p(x), _p(out), _p(cos64.contiguous()),
                                        _p(sin64.contiguous()), B, S, H, *_bshd_strides(x), *_bshd_strides(out),
                                        int(s_rope), codes[x.dtype], codes[out_dtype]), "jenga_rope_complex")
    return out


def wan_norm_rope(x, weight, cos64, sin64, s_rope, eps, out=None):
    """Wan q / k prologue in one pass: x [..., C] (bf16/fp16, uniform row stride) -> bf16 [..., C] = bf16( rope64(
    round(x * rsqrt(mean x^2 + eps)) * weight_fp32 ) ); cos64/sin64 float64 [>= s_rope, 64] or None (norm + cast only).
    `out` may be the first rows of a larger (block-padded) buffer."""
    _need_gpu(x, "wan_norm_rope")
    C = x.shape[-1]
    if cos64 is not None and x.dim() > 2 and x.shape[:-2].numel() != 1:
        # the kernel takes the flattened ROW index as the RoPE position: with a batch > 1 only the first s_rope rows would
        # be rotated, each at the wrong position (the WanSelfAttention call site runs batch 1)
        raise ValueError("wan_norm_rope: with RoPE tables the input must be one sequence ([S, C] or [1, S, C])")
    x2 = x.reshape(-1, C)
    if x2.stride(-1) != 1:
        x2 = x2.contiguous()
    rows = x2.shape[0]
    w = weight.to(device=x.device, dtype=torch.float32).contiguous()
    if out is None:
        out = torch.empty(tuple(x.shape[:-1]) + (C,), dtype=torch.bfloat16, device=x.device)
    o2 = out.reshape(-1, C) if out.is_contiguous() else out
    if o2.dim() != 2 or o2.shape[0] < rows or o2.stride(1) != 1 or out.dtype != torch.bfloat16:
        raise ValueError("wan_norm_rope: out must be bf16 [>= rows, C] with contiguous channels")
    if cos64 is not None:
        if cos64.dtype != torch.float64 or sin64.dtype != torch.float64 or cos64.shape[-1] != 64 or cos64.shape[0] < s_rope:
            raise ValueError("wan_norm_rope: float64 [S,64] tables required")
        cos64, sin64 = cos64.contiguous(), sin64.contiguous()
    else:
        s_rope = 0
    with _on(x.device):
        _check(lib().jenga_wan_norm_rope(_stream(x.device), _p(x2), _p(o2), _p(w), _p(cos64), _p(sin64), rows, C,
                                         x2.stride(0), o2.stride(0), int(s_rope), float(eps), dtype_code(x.dtype)),
               "jenga_wan_norm_rope")
    return out


def _rows2d(t):
    """[..., C] with unit inner stride and a single uniform row stride -> (rows, C, row_stride)."""
    C = t.shape[-1]
    t2 = t.reshape(-1, C) if t.is_contiguous() else t
    if t2.dim() == 3 and t2.shape[0] == 1:
        t2 = t2[0]
    if t2.dim() != 2 or t2.stride(1) != 1:
        raise ValueError("expected a [rows, C] (or [1, rows, C]) tensor with contiguous channels")
    return t2, t2.shape[0], C, t2.stride(0)


def ln_modulate(x, shift, scale, eps=1e-6, shift2=None, scale2=None, mask=None, out=None):
    """x [1,S,C]; shift/scale [1,C] or [C] -> LayerNorm(x)*(1+scale)+shift; rows with mask use (shift2, scale2)."""
    _need_gpu(x, "ln_modulate")
    x2, rows, C, xrs = _rows2d(x)
    if out is None:
        out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    o2, _, _, ors = _rows2d(out)
    v = lambda t: None if t is None else t.reshape(-1).to(dtype=x.dtype).contiguous()
    sh, sc, sh2, sc2 = v(shift), v(scale), v(shift2), v(scale2)
    m = None if mask is None else mask.to(torch.uint8).contiguous()
    with _on(x.device):
        _check(lib().jenga_ln_modulate(_stream(x.device), _p(x2), _p(o2), _p(sh), _p(sc), _p(sh2), _p(sc2), _p(m),
                                       rows, C, xrs, ors, float(eps), dtype_code(x.dtype)), "jenga_ln_modulate")
    return out


def gate_residual(res, y, gate, gate2=None, mask=None, out=None):
    """res + y * gate (per-channel gate [1,C]); rows with mask use gate2."""
    _need_gpu(res, "gate_residual")
    r2, rows, C, rrs = _rows2d(res)
    y2, _, _, yrs = _rows2d(y)
    if out is None:
        out = torch.empty(res.shape, dtype=res.dtype, device=res.device)
    o2, _, _, ors = _rows2d(out)
    g = gate.reshape(-1).to(dtype=res.dtype).contiguous()
    g2 = None if gate2 is None else gate2.reshape(-1).to(dtype=res.dtype).contiguous()
    m = None if mask is None else mask.to(torch.uint8).contiguous()
    with _on(res.device):
        _check(lib().jenga_gate_residual(_stream(res.device), _p(r2), _p(y2), _p(g), _p(g2), _p(m), _p(o2), rows, C,
                                         rrs, yrs, ors, dtype_code(res.dtype)), "jenga_gate_residual")
    return out


def gelu_tanh(x, out=None):
    """tanh-GELU; x / out may be strided views (uniform row stride, contiguous channels)."""
    _need_gpu(x, "gelu_tanh")
    x2, rows, C, xrs = _rows2d(x)
    if out is None:
        out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    o2, _, _, ors = _rows2d(out)
    with _on(x.device):
        _check(lib().jenga_gelu_tanh(_stream(x.device), _p(x2), _p(o2), rows, C, xrs, ors, dtype_code(x.dtype)),
               "jenga_gelu_tanh")
    return out


ACT_NONE, ACT_GELU_TANH = 0, 1
BIAS_F32 = 256
OUT_F32 = 512      # OR-ed into jenga_linear's `act`: the bias vector is float32
_GEMM_WORKSPACE = __import__("collections").OrderedDict()
_GEMM_WORKSPACE_MAX = 8         # scratch buffers kept per process (64 MiB each), least recently used evicted


def _gemm_workspace(device):
    """One 64 MiB scratch buffer per (device, stream): two jenga_linear calls in flight on different streams (the
    sequence-parallel blocks issue GEMMs beside the exchange stream) must not share split-K scratch.  Bounded: at most
    _GEMM_WORKSPACE_MAX buffers are kept (threads simulating ranks, short-lived side streams); an evicted buffer goes back
    to torch's caching allocator, which hands it out again in stream order of the stream it was allocated on -- the stream
    that used it -- so a GEMM still in flight on it is not disturbed."""
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    buf = _GEMM_WORKSPACE.get(key)
    if buf is None:
        buf = _GEMM_WORKSPACE[key] = torch.empty(64 << 20, dtype=torch.uint8, device=device)
        while len(_GEMM_WORKSPACE) > _GEMM_WORKSPACE_MAX:
            _GEMM_WORKSPACE.popitem(last=False)
    else:
        _GEMM_WORKSPACE.move_to_end(key)
    return buf


def linear_export_choices():
    """The current device's jenga_linear algorithm choices as an int64 tensor [n, 12] (CPU)."""
    n = int(lib().jenga_linear_export_choices(None, 0))
    rec = torch.zeros((max(n, 1), 12), dtype=torch.int64)
    n2 = int(lib().jenga_linear_export_choices(ctypes.c_void_p(rec.data_ptr()), n))
    return rec[: min(n, n2)]


def linear_import_choices(records):
    """records int64 [n, 12] (linear_export_choices of the rank whose choices everybody adopts)."""
    rec = records.to(device="cpu", dtype=torch.int64).contiguous()
    if rec.dim() != 2 or rec.shape[1] != 12:
        raise ValueError("linear_import_choices: records must be int64 [n, 12]")
    _check(lib().jenga_linear_import_choices(ctypes.c_void_p(rec.data_ptr()), rec.shape[0]), "jenga_linear_import_choices")


def linear_import_mismatches():
    """Plans rebuilt from imported choices so far whose hipBLASLt solution index is not the exported one (0 = this rank runs
    the exporting rank's algorithms)."""
    return int(lib().jenga_linear_import_mismatches())


def linear(x, weight, bias=None, act=ACT_NONE, gate=None, res=None, out=None):
    """out = act(gate * (x @ weight.T) + bias + res) in ONE hipBLASLt call (jenga_linear): x [..., K] with a uniform
    row stride, weight [N, K] (nn.Linear layout), bias [N] (added as given; a float32 bias is passed on as float32), gate [N] or [1, N] (per-channel; the
    caller folds it into the bias), res / out [..., N] (may be strided views: the MLP half of the single-stream blocks'
    concat buffer).  fp32 accumulation, one rounding."""
    _need_gpu(x, "linear")
    x2, M, K, xrs = _rows2d(x)
    N = weight.shape[0]
    if weight.dim() != 2 or weight.shape[1] != K or weight.stride(1) != 1 or weight.dtype != x.dtype:
        raise ValueError("linear: weight must be [N, K] in the input dtype with contiguous rows")
    # a float32 residual (or out) selects the fp32 C / D form (JENGA_OUT_F32: the Wan blocks' residual stream)
    out32 = (res is not None and res.dtype == torch.float32) or (out is not None and out.dtype == torch.float32)
    odt = torch.float32 if out32 else x.dtype
    if out is None:
        out = torch.empty(tuple(x.shape[:-1]) + (N,), dtype=odt, device=x.device)
    o2, Mo, No, ors = _rows2d(out)
    if Mo != M or No != N or out.dtype != odt:
        raise ValueError("linear: out must be [..., N] with as many rows as x (float32 when the residual is float32)")
    r2, rrs = None, 0
    if res is not None:
        r2, Mr, Nr, rrs = _rows2d(res)
        if Mr != M or Nr != N or res.dtype != odt:
            raise ValueError("linear: res must match out")
    if out32:
        if int(act) & 0xff:
            raise ValueError("linear: the float32 residual form has no activation epilogue")
        act = int(act) | OUT_F32
    if bias is not None:
        if bias.dtype == torch.float32:     # (the gated bias of proj / fc2 / linear2: kept unrounded)
            act = int(act) | BIAS_F32
            bias = bias.reshape(-1).contiguous()
        else:
            bias = bias.reshape(-1).to(dtype=x.dtype).contiguous()
        if bias.numel() != N:
            raise ValueError("linear: bias must have N entries")
    g32 = None
    if gate is not None:
        g32 = gate.reshape(-1).to(dtype=torch.float32).contiguous()
        if g32.numel() != N:
            raise ValueError("linear: gate must have N entries")
    ws = _gemm_workspace(x.device)
    with _on(x.device):
        _check(lib().jenga_linear(_stream(x.device), _p(x2), _p(weight), _p(bias), _p(r2), _p(g32), _p(o2), M, N, K, xrs,
                                  weight.stride(0), rrs, ors, int(act), _p(ws), ws.numel(), dtype_code(x.dtype)),
               "jenga_linear")
    return out


def wan_ln_modulate(x, weight=None, bias=None, shift=None, scale=None, eps=1e-6, out_dtype=torch.bfloat16,
                    round_ln=False):
    """Wan block glue: x fp32 [1,S,C] -> out_dtype( LN(x) [*weight+bias] [*(1+scale)+shift] ), vectors fp32 [C]."""
    _need_gpu(x, "wan_ln_modulate")
    if x.dtype != torch.float32:
        raise ValueError("wan_ln_modulate: the Wan residual stream is float32")
    x2, rows, C, xrs = _rows2d(x)
    out = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    o2, _, _, ors = _rows2d(out)
    v = lambda t: None if t is None else t.reshape(-1).to(dtype=torch.float32).contiguous()
    w, b, sh, sc = v(weight), v(bias), v(shift), v(scale)
    with _on(x.device):
        _check(lib().jenga_wan_ln_modulate(_stream(x.device), _p(x2), _p(o2), _p(w), _p(b), _p(sh), _p(sc), rows, C,
                                           xrs, ors, float(eps), dtype_code(out_dtype), int(bool(round_ln))),
               "jenga_wan_ln_modulate")
    return out


def wan_gate_residual(x, y, gate=None, out=None):
    """x fp32 + float(y) [* gate] -> fp32; out=x updates the residual stream in place."""
    _need_gpu(x, "wan_gate_residual")
    if x.dtype != torch.float32:
        raise ValueError("wan_gate_residual: the Wan residual stream is float32")
    x2, rows, C, xrs = _rows2d(x)
    y2, _, _, yrs = _rows2d(y)
    if out is None:
        out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    o2, _, _, ors = _rows2d(out)
    g = None if gate is None else gate.reshape(-1).to(dtype=torch.float32).contiguous()
    with _on(x.device):
        _check(lib().jenga_wan_gate_residual(_stream(x.device), _p(x2), _p(y2), _p(g), _p(o2), rows, C, xrs, yrs, ors,
                                             dtype_code(y.dtype)), "jenga_wan_gate_residual")
    return out


def block_pool(x, n_blocks):
    """x [B,S,H,128] -> [B,H,n_blocks,128] means over 128-token blocks."""
    _need_gpu(x, "block_pool")
    B, S, H, D = x.shape
    if D != 128 or n_blocks * 128 > S:
        raise ValueError("block_pool: head_dim must be 128 and n_blocks*128 <= S")
    out = torch.empty((B, H, n_blocks, 128), dtype=x.dtype, device=x.device)
    with _on(x.device):
        _check(lib().jenga_block_pool(_stream(x.device), _p(x), _p(out), B, H, n_blocks, *_bshd_strides(x),
                                      dtype_code(x.dtype)), "jenga_block_pool")
    return out


def block_select(qpool, kpool, neighbors, nk_img, text_blocks, top_k, p, first_frame_blocks=0, want_mask=False,
                 want_lists=True, flags=None, head_dim=128):
    """-> (mask uint8 [B,H,nq,nk_all] | None, idx int32 [B,H,nq,nk_all] | None, cnt int32 [B,H,nq] | None).
    flags: SELECT_DEVICE_SCAN = the kept-count rule with torch's DEVICE cumsum semantics (default: CPU semantics, or
    the JENGA_SELECT_FLAGS environment variable).  head_dim in (16, 32, 64): the pooled rows are zero-padded to 128
    channels and the scores are scaled by head_dim ** -0.5 (JENGA_SELECT_HEAD_DIM)."""
    _need_gpu(qpool, "block_select")
    if head_dim not in (16, 32, 64, 128):
        raise ValueError(f"block_select: head_dim must be 16, 32, 64 or 128 (got {head_dim})")
    B, H, nq, _ = qpool.shape
    nk_all = nk_img + text_blocks
    if kpool.shape != (B, H, nk_all, 128):
        raise ValueError(f"kpool shape {tuple(kpool.shape)} != {(B, H, nk_all, 128)}")
    dev = qpool.device
    mask = torch.empty((B, H, nq, nk_all), dtype=torch.uint8, device=dev) if want_mask else None
    idx = torch.empty((B, H, nq, nk_all), dtype=torch.int32, device=dev) if want_lists else None
    cnt = torch.empty((B, H, nq), dtype=torch.int32, device=dev) if want_lists else None
    nbr, nbc = 0, 0
    if neighbors is not None:
        if neighbors.dtype == torch.bool:
            neighbors = neighbors.view(torch.uint8)
        neighbors = neighbors.to(dev).contiguous()
        nbr, nbc = neighbors.shape
    with _on(dev):
        _check(lib().jenga_block_select(_stream(dev), _p(qpool.contiguous()), _p(kpool.contiguous()), _p(neighbors),
                                        nbr, nbc, _p(mask), _p(idx), _p(cnt), B, H, nq, nk_img, text_blocks,
                                        int(top_k), float(p), int(first_frame_blocks), dtype_code(qpool.dtype),
                                        int(SELECT_DEFAULT_FLAGS if flags is None else flags) |
                                        (0 if head_dim == 128 else (head_dim << 8))),
               "jenga_block_select")
    return mask, idx, cnt


def pack_v(v, n_blocks=None, out=None, dst_block0=0, dst_blocks_total=None):
    """v [B,S,H,128] -> opaque re-tiled workspace [B,H,2*blocks_total,128,64] for bsattn_fwd.
    With out/dst_block0/dst_blocks_total a part (e.g. image or text V) fills its slice of a shared workspace."""
    _need_gpu(v, "pack_v")
    B, S, H, D = v.shape
    if n_blocks is None:
        n_blocks = S // 128
    if D != 128 or n_blocks * 128 != S:
        raise ValueError("pack_v: S must equal n_blocks*128 and head_dim 128")
    if dst_blocks_total is None:
        dst_blocks_total = n_blocks if out is None else out.shape[2] // 2
    if out is None:
        out = torch.empty((B, H, dst_blocks_total * 2, 128, 64), dtype=v.dtype, device=v.device)
    with _on(v.device):
        _check(lib().jenga_pack_v(_stream(v.device), _p(v), _p(out), B, H, n_blocks, *_bshd_strides(v),
                                  dst_block0, dst_blocks_total, dtype_code(v.dtype)), "jenga_pack_v")
    return out


def bsattn_fwd(q, k, vt, seqlens, idx, cnt, nq_img, sm_scale, text_amp, text_block_start, out=None, xcd_remap=True,
               flags=None, order=None):
    """q,k [B,S,H,128]; vt from pack_v; seqlens int32 [B] device; idx/cnt from block_select -> o [B,S,H,128].
    order: optional launch-order hint (order_by_count); flags & ATTN_SORTED builds it from cnt."""
    _need_gpu(q, "bsattn_fwd")
    B, S, H, D = q.shape
    if D != 128 or S % 128:
        raise ValueError("bsattn_fwd: head_dim must be 128 and S a multiple of 128")
    n_blocks = S // 128
    if out is None:
        out = torch.empty((B, S, H, 128), dtype=q.dtype, device=q.device)
    if seqlens is not None and (seqlens.dtype != torch.int32 or seqlens.device != q.device):
        seqlens = seqlens.to(device=q.device, dtype=torch.int32)
    if seqlens is not None and seqlens.numel() < B:
        # the reference passes cu_seqlens_q[1:2] whatever the batch (attention_block_triton_diffres.py:327-329) and its
        # kernel then reads past it for B > 1; here that is an error instead of an out-of-bounds read
        raise ValueError(f"bsattn_fwd: seqlens has {seqlens.numel()} entries for batch {B}")
    if idx is not None and idx.shape[-1] != n_blocks:
        raise ValueError("idx row length must equal the number of kv blocks")
    if k.shape != q.shape or out.shape != q.shape:
        raise ValueError("bsattn_fwd: q, k and out must have the same [B,S,H,128] shape")
    if tuple(vt.shape) != (B, H, 2 * n_blocks, 128, 64) or not vt.is_contiguous() or vt.dtype != q.dtype:
        raise ValueError(f"bsattn_fwd: vt must be the contiguous pack_v workspace {(B, H, 2 * n_blocks, 128, 64)}")
    if nq_img > 0 and (idx is None or cnt is None or tuple(idx.shape[:3]) != (B, H, nq_img)
                       or tuple(cnt.shape) != (B, H, nq_img) or idx.dtype != torch.int32 or cnt.dtype != torch.int32
                       or not idx.is_contiguous() or not cnt.is_contiguous()):
        raise ValueError("bsattn_fwd: idx / cnt must be contiguous int32 [B,H,nq_img,n_blocks] / [B,H,nq_img]")
    fl = ATTN_DEFAULT_FLAGS if flags is None else flags
    if not xcd_remap:
        fl &= ~ATTN_XCD_REMAP
    pair = bool(fl & ATTN_PAIR)
    if pair and order is not None:
        # `order` ranks query BLOCKS; the pair kernel launches block PAIRS in an order of its own (ATTN_SORTED) -- dropping a
        # caller's hint silently would misreport what ran
        raise ValueError("bsattn_fwd: a launch-order hint (order=) cannot be combined with the pair kernel (ATTN_PAIR)")
    prof = ATTN_PROFILE
    with _on(q.device):
        pidx = pcnt = order_t = None
        porder = None
        if pair and nq_img > 0:
            pidx, pcnt = pair_merge(idx, cnt, n_blocks)
            npair = pidx.shape[2]
            if (fl & ATTN_SORTED) and npair > 1:
                # work-aware launch order of the pairs inside every XCD's range: a shared block costs two 32-row items per
                # wave and tile, an unshared one one
                seg = (npair + 7) // 8 if ((fl & ATTN_XCD_REMAP) and npair >= 64) else npair
                if seg <= 2048:
                    work = (2 * pcnt[..., 0] + pcnt[..., 1] + pcnt[..., 2]).contiguous()
                    porder = order_by_count(work, seg)
        if order is not None:
            order_t = order
        elif (fl & ATTN_SORTED) and nq_img > 0 and (fl & ATTN_LP) and not pair:
            seg = (nq_img + 7) // 8 if ((fl & ATTN_XCD_REMAP) and nq_img >= 64) else nq_img
            if seg <= 2048:      # (jenga_order_by_count ranks one segment per workgroup; beyond that: plain order)
                order_t = order_by_count(cnt, seg)
        if order_t is not None and (tuple(order_t.shape) != (B, H, nq_img) or order_t.dtype != torch.int32
                                    or not order_t.is_contiguous() or order_t.device != q.device):
            raise ValueError("bsattn_fwd: order must be a contiguous int32 [B,H,nq_img] tensor on the device")
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        common = (B, H, n_blocks, nq_img, *_bshd_strides(q), *_bshd_strides(k), *_bshd_strides(out), float(sm_scale),
                  float(text_amp), int(text_block_start), dtype_code(q.dtype))
        cflags = fl & (ATTN_XCD_REMAP | ATTN_BALANCE | ATTN_LP)
        if not pair:
            _check(lib().jenga_bsattn_fwd(_stream(q.device), _p(q), _p(k), _p(vt), _p(out), _p(seqlens), _p(idx),
                                          _p(cnt), _p(order_t), *common, cflags), "jenga_bsattn_fwd")
        else:
            _check(lib().jenga_bsattn_pair_fwd(_stream(q.device), _p(q), _p(k), _p(vt), _p(out), _p(seqlens),
                                               _p(pidx), _p(pcnt), _p(porder), *common, cflags & ~ATTN_LP),
                   "jenga_bsattn_pair_fwd")
        if prof is not None:
            e1.record()
            prof.events.append((e0, e1))
            prof.launches += 1
            pairs = B * H * (n_blocks - nq_img) * n_blocks
            tot = (cnt.sum(dtype=torch.int64) if cnt is not None else 0) + pairs
            prof.pairs = tot if prof.pairs is None else prof.pairs + tot
            prof.per_launch.append((prof.tag, tot))
            if idx is not None and nq_img > 1:
                prof.last_lists = (idx, cnt)
    return out


def cross_attn_fwd(q, k, v, sm_scale=None, out=None, kv_len=None):
    """Dense cross-attention (WanT2VCrossAttention): q [B,Sq,H,128], k / v [B,Skv,H,128], Sq and Skv multiples of 128
    (pad the buffers; query rows are independent, keys >= kv_len are masked) -> o [B,Sq,H,128]."""
    _need_gpu(q, "cross_attn_fwd")
    B, Sq, H, D = q.shape
    Skv = k.shape[1]
    if D != 128 or Sq % 128 or Skv % 128 or Sq == 0 or Skv == 0 or tuple(k.shape) != (B, Skv, H, 128) \
            or v.shape != k.shape or k.dtype != q.dtype or v.dtype != q.dtype:
        raise ValueError("cross_attn_fwd: q [B,Sq,H,128], k / v [B,Skv,H,128] of one dtype, Sq and Skv multiples of 128")
    kv_len = Skv if kv_len is None else int(kv_len)
    if not (Skv - 128 < kv_len <= Skv):
        raise ValueError("cross_attn_fwd: kv_len must lie inside the last 128-key block of k / v")
    if out is None:
        out = torch.empty((B, Sq, H, 128), dtype=q.dtype, device=q.device)
    vt = pack_v(v, Skv // 128)
    with _on(q.device):
        _check(lib().jenga_cross_attn_fwd(_stream(q.device), _p(q), _p(k), _p(vt), _p(out), B, H, Sq // 128, Skv // 128,
                                          kv_len, *_bshd_strides(q), *_bshd_strides(k), *_bshd_strides(out),
                                          float(D ** -0.5 if sm_scale is None else sm_scale), dtype_code(q.dtype)),
               "jenga_cross_attn_fwd")
    return out


def order_by_count(cnt, segment):
    """cnt int32 [B,H,nq] -> order int32 [B,H,nq]: inside every run of `segment` consecutive query blocks, the blocks by
    descending kept count (the launch-order hint of jenga_bsattn_fwd)."""
    _need_gpu(cnt, "order_by_count")
    if cnt.dim() != 3 or cnt.dtype != torch.int32 or not cnt.is_contiguous():
        raise ValueError("order_by_count: cnt must be a contiguous int32 [B,H,nq] tensor")
    B, H, nq = cnt.shape
    order = torch.empty_like(cnt)
    with _on(cnt.device):
        _check(lib().jenga_order_by_count(_stream(cnt.device), _p(cnt), B * H, nq, int(segment), _p(order)),
               "jenga_order_by_count")
    return order


def pair_merge(idx, cnt, n_blocks):
    """idx int32 [B,H,nq,n_blocks], cnt int32 [B,H,nq] (jenga_block_select) -> (pidx
    [B,H,ceil(nq/2),n_blocks], pcnt [B,H,ceil(nq/2),4]): per query-block pair the kv blocks both keep | only the even
    row | only the odd row."""
    _need_gpu(idx, "pair_merge")
    B, H, nq, nb = idx.shape
    if nb != n_blocks or tuple(cnt.shape) != (B, H, nq) or idx.dtype != torch.int32 or cnt.dtype != torch.int32:
        raise ValueError("pair_merge: idx / cnt must be int32 [B,H,nq,n_blocks] / [B,H,nq]")
    npair = (nq + 1) // 2
    pidx = torch.empty((B, H, npair, nb), dtype=torch.int32, device=idx.device)
    pcnt = torch.empty((B, H, npair, 4), dtype=torch.int32, device=idx.device)
    with _on(idx.device):
        _check(lib().jenga_pair_merge(_stream(idx.device), _p(idx.contiguous()), _p(cnt.contiguous()), B, H, nq, nb,
                                      _p(pidx), _p(pcnt)), "jenga_pair_merge")
    return pidx, pcnt


def ulysses_pack_heads(x, n_ranks, out=None):
    """x [B,S_loc,H,128] -> [N,B,S_loc,H/N,128] (peer-major send buffer)."""
    _need_gpu(x, "ulysses_pack_heads")
    B, S, H, D = x.shape
    if out is None:
        out = torch.empty((n_ranks, B, S, H // n_ranks, D), dtype=x.dtype, device=x.device)
    with _on(x.device):
        _check(lib().jenga_ulysses_pack_heads(_stream(x.device), _p(x), _p(out), B, S, H, n_ranks,
                                              *_bshd_strides(x)), "jenga_ulysses_pack_heads")
    return out


def ulysses_unpack_heads(recv, n_ranks, out=None):
    """recv [N,B,S_loc,H/N,128] -> [B,S_loc,H,128]."""
    _need_gpu(recv, "ulysses_unpack_heads")
    N, B, S, Hn, D = recv.shape
    if out is None:
        out = torch.empty((B, S, Hn * N, D), dtype=recv.dtype, device=recv.device)
    with _on(recv.device):
        _check(lib().jenga_ulysses_unpack_heads(_stream(recv.device), _p(recv.contiguous()), _p(out), B, S, Hn * N, N,
                                                *_bshd_strides(out)), "jenga_ulysses_unpack_heads")
    return out
