// HBM-bound row kernels: Hilbert gather/scatter, fused per-head RMSNorm + RoPE, 128-token block pooling,
// V re-tiling for the P.V product, Ulysses head pack/unpack.  All move 16 bytes per lane per access.
// Compiled with -ffp-contract=off: the reference's eager fp32 arithmetic has no fused multiply-adds.
#include "common.h"

namespace jenga {
namespace {

// Streaming accesses (round 4): the row kernels read every input byte and write every output byte once, hundreds of MB per
// launch -- nontemporal loads / stores keep them from rotating through the L2 / Infinity Cache (tools/micro/hbm_copy.hip:
// a 2 x 708 MB copy runs at 6.58 TB/s with them, 6.16 without; at a capped grid 5.75 vs 4.96): gather 5.5 -> 6.3 TB/s,
// pack_v 5.2-5.5 -> 6.0, LayerNorm+modulate 5.5 -> 5.6 (profiles/r04_row_kernels_streaming_ab.json).  Small reused operands
// (weights, modulation rows, cos / sin tables, indices) keep ordinary loads.
typedef unsigned int jenga_u32x4 __attribute__((ext_vector_type(4)));
#ifdef JENGA_NO_NT      // A/B switch (tools/diag_det.py): ordinary accesses instead of the streaming ones
__device__ __forceinline__ uint4 ld_stream(const void* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ void st_stream(void* p, const uint4 v) { *reinterpret_cast<uint4*>(p) = v; }
#else
__device__ __forceinline__ uint4 ld_stream(const void* p) {
    const jenga_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const jenga_u32x4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st_stream(void* p, const uint4 v) {
    const jenga_u32x4 w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<jenga_u32x4*>(p));
}
#endif

// ------------------------------------------------------------------------------------------------ gather
// dst[b, i, :] = src[b, index[i], :]; one workgroup per output row (grid-stride), 16 B per lane.
__global__ void gather_rows_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst,
                                   const int64_t* __restrict__ index, long long batch, long long n_rows,
                                   int vec_per_row, long long src_bs, long long dst_bs) {
    const long long total = batch * n_rows;
    for (long long r = blockIdx.x; r < total; r += gridDim.x) {
        const long long b = r / n_rows, i = r % n_rows;
        const long long s = index[i];
        const uint4* sp = src + b * src_bs + s * vec_per_row;
        uint4* dp = dst + b * dst_bs + i * vec_per_row;
        for (int v = threadIdx.x; v < vec_per_row; v += blockDim.x) st_stream(dp + v, ld_stream(sp + v));
    }
}

// ------------------------------------------------------------------------------------------------ RMSNorm + RoPE
// 16 lanes own one (token, head) row of 128 elements (8 each); a wave covers 4 heads of one token, a 256-thread
// workgroup 16 heads; the cos/sin row of the token is shared by all heads (L1/L2 hits).
template <typename T>
__global__ void rmsnorm_rope_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ out,
                                    const uint16_t* __restrict__ weight, const float* __restrict__ cosT,
                                    const float* __restrict__ sinT, long long B, long long S, long long H,
                                    long long x_sb, long long x_ss, long long x_sh, long long o_sb, long long o_ss,
                                    long long o_sh, long long s_rope, float eps) {
    const int sub = threadIdx.x & 15;          // 8-element slice of the row
    const int rig = threadIdx.x >> 4;          // row-in-group (0..15)
    const long long rows = B * S * H;
    float wv[8];
    if (weight) {
        unpack8<T>(*reinterpret_cast<const uint4*>(weight + sub * 8), wv);
    }
    for (long long row = (long long)blockIdx.x * 16 + rig; row < rows; row += (long long)gridDim.x * 16) {
        const long long h = row % H, s = (row / H) % S, b = row / (H * S);
        float f[8];
        unpack8<T>(*reinterpret_cast<const uint4*>(x + b * x_sb + s * x_ss + h * x_sh + sub * 8), f);
        if (eps >= 0.f) {  // eps < 0: RoPE only (apply_rotary_emb on already-normalised tensors)
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) ss = __fadd_rn(ss, __fmul_rn(f[i], f[i]));
            ss += __shfl_xor(ss, 1);
            ss += __shfl_xor(ss, 2);
            ss += __shfl_xor(ss, 4);
            ss += __shfl_xor(ss, 8);
            const float r = __fdiv_rn(1.0f, __fsqrt_rn(__fadd_rn(__fdiv_rn(ss, 128.0f), eps)));
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float y = round_to<T>(__fmul_rn(f[i], r));                // ._norm(x.float()).type_as(x)
                if (weight) y = round_to<T>(__fmul_rn(y, wv[i]));        // * weight (dtype * dtype -> dtype)
                f[i] = y;
            }
        }
        if (cosT && s < s_rope) {
            const float4* cp = reinterpret_cast<const float4*>(cosT + s * 128 + sub * 8);
            const float4* sp = reinterpret_cast<const float4*>(sinT + s * 128 + sub * 8);
            const float4 c0 = cp[0], c1 = cp[1], s0 = sp[0], s1 = sp[1];
            const float c[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
            const float sn[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
            float g[8];
#pragma unroll
            for (int i = 0; i < 8; i += 2) {  // rotate_half: (x0,x1) -> (-x1, x0)
                g[i] = __fadd_rn(__fmul_rn(f[i], c[i]), __fmul_rn(-f[i + 1], sn[i]));
                g[i + 1] = __fadd_rn(__fmul_rn(f[i + 1], c[i + 1]), __fmul_rn(f[i], sn[i + 1]));
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] = g[i];
        }
        *reinterpret_cast<uint4*>(out + b * o_sb + s * o_ss + h * o_sh + sub * 8) = pack8<T>(f);
    }
}

// ------------------------------------------------------------------------------------------------ Q+K norm+RoPE+pool
// the RMSNorm / RoPE pass and the two block-pooling passes of a layer in ONE kernel.  One workgroup
// per 128-token block, 16 lanes per head (blockDim = 16 H): a thread keeps its 8-element slice of ONE head for the
// whole block, so the pooled mean of Q and K is accumulated in registers while the rows stream through -- in exactly
// the order of block_pool_kernel (8 tokens summed, 16 partial sums added in ascending order, one rounding), on the
// values already rounded to the storage dtype, i.e. pooled outputs are bit-identical to pooling the written tensors.
// The cos / sin row of a token is fetched once for Q and K and all heads (rmsnorm_rope_kernel, called per tensor,
// fetched the tables twice per layer).  Arithmetic of the norm and the rotation: rmsnorm_rope_kernel's, verbatim.
// x + its three xor-partners inside a row of 16 lanes, in the order of the __shfl_xor(1, 2, 4, 8) butterfly, as four DPP adds
// (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror) instead of four ds_bpermute round trips through the
// LDS crossbar.  Same operands in every add (after the first two steps the lanes of a quad hold one value, so "the mirrored lane"
// and "the lane 4 / 8 away" carry the same number; fp32 addition commutes): bit-identical (round 6; tools/bench_rowops.py
// compares the checksums of both builds at the full shape).
__device__ __forceinline__ float row16_allsum(float x) {
#define JENGA_DPP_ADD(CTRL_)                                                                                              \
    x = __fadd_rn(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), (CTRL_), 0xf, 0xf, \
                                                                           false)))
    JENGA_DPP_ADD(0xB1);
    JENGA_DPP_ADD(0x4E);
    JENGA_DPP_ADD(0x141);
    JENGA_DPP_ADD(0x140);
#undef JENGA_DPP_ADD
    return x;
}
// a and b rounded to T and back, through ONE packed conversion (v_cvt_pk_bf16_f32 converts two values; written per element the
// compiler spends one conversion and one shift on each)
template <typename T>
__device__ __forceinline__ void round_to2(float& a, float& b) {
    const uint32_t w = pack2<T>(a, b);
    a = to_f32<T>((uint16_t)(w & 0xffffu));
    b = to_f32<T>((uint16_t)(w >> 16));
}

template <typename T>
__device__ __forceinline__ void norm_rope_row(float (&f)[8], const float (&wv)[8], bool has_w, float eps, bool rope,
                                              const float (&c)[8], const float (&sn)[8], uint4* packed = nullptr) {
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) ss = __fadd_rn(ss, __fmul_rn(f[i], f[i]));
    ss = row16_allsum(ss);
    const float r = __fdiv_rn(1.0f, __fsqrt_rn(__fadd_rn(__fdiv_rn(ss, 128.0f), eps)));
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        float y0 = __fmul_rn(f[i], r), y1 = __fmul_rn(f[i + 1], r);
        round_to2<T>(y0, y1);
        if (has_w) {
            y0 = __fmul_rn(y0, wv[i]);
            y1 = __fmul_rn(y1, wv[i + 1]);
            round_to2<T>(y0, y1);
        }
        f[i] = y0;
        f[i + 1] = y1;
    }
    if (rope) {
        float g[8];
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            g[i] = __fadd_rn(__fmul_rn(f[i], c[i]), __fmul_rn(-f[i + 1], sn[i]));
            g[i + 1] = __fadd_rn(__fmul_rn(f[i + 1], c[i + 1]), __fmul_rn(f[i], sn[i + 1]));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = g[i];
    }
    // what the store writes (and what gets pooled): rounded pair by pair; the packed words ARE the store's payload
    uint32_t pw[4];
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        pw[i >> 1] = pack2<T>(f[i], f[i + 1]);
        f[i] = to_f32<T>((uint16_t)(pw[i >> 1] & 0xffffu));
        f[i + 1] = to_f32<T>((uint16_t)(pw[i >> 1] >> 16));
    }
    if (packed) *packed = make_uint4(pw[0], pw[1], pw[2], pw[3]);
}

template <typename T>
__global__ void __launch_bounds__(128) qk_norm_rope_pool_kernel(const uint16_t* __restrict__ xq, const uint16_t* __restrict__ xk,
                                         uint16_t* __restrict__ oq, uint16_t* __restrict__ ok,
                                         const uint16_t* __restrict__ wq, const uint16_t* __restrict__ wk,
                                         const float* __restrict__ cosT, const float* __restrict__ sinT,
                                         uint16_t* __restrict__ qpool, uint16_t* __restrict__ kpool, long long B,
                                         long long nblk, long long H, long long x_sb, long long x_ss, long long x_sh,
                                         long long o_sb, long long o_ss, long long o_sh, long long s_rope,
                                         long long pool_block0, long long nq_pool, long long nk_pool, float eps,
                                         int HG) {
    // blockDim.x == 16 * HG: a workgroup owns HG heads of one 128-token block; the H / HG workgroups of a block sit
    // on one XCD next to each other in launch order (they share the block's cos / sin rows through that L2)
    const int sub = threadIdx.x & 15;
    const int ngrp = (int)(H / HG);
    float wqv[8], wkv[8];
    if (wq) unpack8<T>(*reinterpret_cast<const uint4*>(wq + sub * 8), wqv);
    if (wk) unpack8<T>(*reinterpret_cast<const uint4*>(wk + sub * 8), wkv);
    for (long long id = blockIdx.x; id < ((B * nblk + 7) / 8) * 8 * ngrp; id += gridDim.x) {
        // id = 8 * (ngrp * (blk / 8) + grp) + blk % 8  ->  the groups of a block share id % 8 (= the XCD)
        const long long blk8 = id / (8 * ngrp), rem = id % (8 * ngrp);
        const long long wg = blk8 * 8 + (rem & 7);
        const long long h = (rem >> 3) * HG + (threadIdx.x >> 4);
        if (wg >= B * nblk) continue;
        const long long j = wg % nblk, b = wg / nblk;
        float sq[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        float aq[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ak[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        // the raw rows of the NEXT token are in flight while this one is normalised (each thread walks 128 tokens;
        // deeper register rings spill at the 128 registers that 1024-thread launch bounds leave, and do not help)
        const long long x0 = b * x_sb + (j * 128) * x_ss + h * x_sh + sub * 8;
        uint4 rq = *reinterpret_cast<const uint4*>(xq + x0), rk = *reinterpret_cast<const uint4*>(xk + x0);
#pragma unroll 1
        for (int t = 0; t < 128; ++t) {
            const long long s_tok = j * 128 + t;
            const uint4 cq = rq, ck = rk;
            if (t + 1 < 128) {
                rq = *reinterpret_cast<const uint4*>(xq + x0 + (long long)(t + 1) * x_ss);
                rk = *reinterpret_cast<const uint4*>(xk + x0 + (long long)(t + 1) * x_ss);
            }
            const long long ooff = b * o_sb + s_tok * o_ss + h * o_sh + sub * 8;
            float fq[8], fk[8];
            unpack8<T>(cq, fq);
            unpack8<T>(ck, fk);
            const bool rope = cosT && s_tok < s_rope;
            float c[8], sn[8];
            if (rope) {
                const float4* cp = reinterpret_cast<const float4*>(cosT + s_tok * 128 + sub * 8);
                const float4* sp = reinterpret_cast<const float4*>(sinT + s_tok * 128 + sub * 8);
                const float4 c0 = cp[0], c1 = cp[1], s0 = sp[0], s1 = sp[1];
                c[0] = c0.x; c[1] = c0.y; c[2] = c0.z; c[3] = c0.w; c[4] = c1.x; c[5] = c1.y; c[6] = c1.z; c[7] = c1.w;
                sn[0] = s0.x; sn[1] = s0.y; sn[2] = s0.z; sn[3] = s0.w; sn[4] = s1.x; sn[5] = s1.y; sn[6] = s1.z; sn[7] = s1.w;
            }
            uint4 pq4, pk4;
            norm_rope_row<T>(fq, wqv, wq != nullptr, eps, rope, c, sn, &pq4);
            norm_rope_row<T>(fk, wkv, wk != nullptr, eps, rope, c, sn, &pk4);
            *reinterpret_cast<uint4*>(oq + ooff) = pq4;      // (streaming accesses measured here: 4.81-4.87 against
            *reinterpret_cast<uint4*>(ok + ooff) = pk4;      //  4.89-5.14 TB/s with ordinary ones -- not used)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                aq[e] += fq[e];
                ak[e] += fk[e];
            }
            if ((t & 7) == 7) {   // block_pool_kernel's order: 8 tokens summed, the 16 partial sums added in order
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    sq[e] += aq[e];
                    sk[e] += ak[e];
                    aq[e] = 0.f;
                    ak[e] = 0.f;
                }
            }
        }
        float pq[8], pk[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            pq[e] = sq[e] / 128.0f;
            pk[e] = sk[e] / 128.0f;
        }
        const long long jp = pool_block0 + j;
        if (qpool && jp < nq_pool) *reinterpret_cast<uint4*>(qpool + ((b * H + h) * nq_pool + jp) * 128 + sub * 8) = pack8<T>(pq);
        if (kpool && jp < nk_pool) *reinterpret_cast<uint4*>(kpool + ((b * H + h) * nk_pool + jp) * 128 + sub * 8) = pack8<T>(pk);
    }
}

// ------------------------------------------------------------------------------------------------ SP prologue
// SURVEY.md §8 f-2 on the sequence-parallel path (models_mul_block_gc_ha_multigpu.py:196-214 feeding
// xdit_ring_atten.py:118-131): ONE pass over the local shard's Q, K and V slices of the QKV GEMM output does the
// per-head RMSNorm + RoPE of Q and K and writes all three straight into the PEER-MAJOR send buffers of the Ulysses
// exchange (head h goes to peer h / Hn as its local head h % Hn) -- what used to be rmsnorm_rope x 2 +
// ulysses_pack_heads x 3.  Any row count (no 128-token blocking: pooling happens after the exchange, on the gathered
// sequence).  With o_sp = 0 and a head window [head0, head0 + n_heads) the same kernel writes a rank's OWN head slice
// of the replicated text rows directly behind the gathered image rows of the attention inputs.
// Arithmetic of the norm and the rotation: norm_rope_row (= rmsnorm_rope_kernel's, bit for bit).
template <typename T>
__global__ void __launch_bounds__(256) sp_qkv_prologue_kernel(
    const uint16_t* __restrict__ xq, const uint16_t* __restrict__ xk, const uint16_t* __restrict__ xv,
    uint16_t* __restrict__ oq, uint16_t* __restrict__ ok, uint16_t* __restrict__ ov, const uint16_t* __restrict__ wq,
    const uint16_t* __restrict__ wk, const float* __restrict__ cosT, const float* __restrict__ sinT, long long B,
    long long S, long long head0, long long n_heads, long long Hn, long long x_sb, long long x_ss, long long x_sh,
    long long o_sp, long long o_sb, long long o_ss, long long o_sh, long long s_rope, float eps) {
    const int sub = threadIdx.x & 15, rig = threadIdx.x >> 4;
    const long long rows = B * S * n_heads;
    float wqv[8], wkv[8];
    if (wq) unpack8<T>(*reinterpret_cast<const uint4*>(wq + sub * 8), wqv);
    if (wk) unpack8<T>(*reinterpret_cast<const uint4*>(wk + sub * 8), wkv);
    for (long long row = (long long)blockIdx.x * 16 + rig; row < rows; row += (long long)gridDim.x * 16) {
        const long long hw = row % n_heads, s = (row / n_heads) % S, b = row / (n_heads * S);
        const long long h = head0 + hw;
        const long long xo = b * x_sb + s * x_ss + h * x_sh + sub * 8;
        const long long oo = (h / Hn) * o_sp + b * o_sb + s * o_ss + (h % Hn) * o_sh + sub * 8;
        // (xq, xk) or xv may be absent: the blocks issue the Q|K and the V GEMMs separately so that the Q, K exchange
        // is already in flight while the V GEMM runs (pointers are launch-uniform: no divergence)
        if (xv) *reinterpret_cast<uint4*>(ov + oo) = *reinterpret_cast<const uint4*>(xv + xo);
        if (!xq) continue;
        const uint4 rq = *reinterpret_cast<const uint4*>(xq + xo);
        const uint4 rk = *reinterpret_cast<const uint4*>(xk + xo);
        float fq[8], fk[8];
        unpack8<T>(rq, fq);
        unpack8<T>(rk, fk);
        const bool rope = cosT && s < s_rope;
        float c[8], sn[8];
        if (rope) {
            const float4* cp = reinterpret_cast<const float4*>(cosT + s * 128 + sub * 8);
            const float4* sp = reinterpret_cast<const float4*>(sinT + s * 128 + sub * 8);
            const float4 c0 = cp[0], c1 = cp[1], s0 = sp[0], s1 = sp[1];
            c[0] = c0.x; c[1] = c0.y; c[2] = c0.z; c[3] = c0.w; c[4] = c1.x; c[5] = c1.y; c[6] = c1.z; c[7] = c1.w;
            sn[0] = s0.x; sn[1] = s0.y; sn[2] = s0.z; sn[3] = s0.w; sn[4] = s1.x; sn[5] = s1.y; sn[6] = s1.z; sn[7] = s1.w;
        }
        norm_rope_row<T>(fq, wqv, wq != nullptr, eps, rope, c, sn);
        norm_rope_row<T>(fk, wkv, wk != nullptr, eps, rope, c, sn);
        *reinterpret_cast<uint4*>(oq + oo) = pack8<T>(fq);
        *reinterpret_cast<uint4*>(ok + oo) = pack8<T>(fk);
    }
}

// ------------------------------------------------------------------------------------------------ block pooling
// One workgroup per (b, h, block): 128 tokens x 128 dims.  thread -> (token group tg = t/16, slice = t%16);
// each thread sums 8 tokens in fp32, LDS tree over the 16 groups, one rounding of mean to dtype.
template <typename T>
__global__ void block_pool_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ pooled, long long B,
                                  long long H, long long nb, long long x_sb, long long x_ss, long long x_sh) {
    __shared__ float red[16][129];
    const int sub = threadIdx.x & 15, tg = threadIdx.x >> 4;
    for (long long blk = blockIdx.x; blk < B * H * nb; blk += gridDim.x) {
        const long long j = blk % nb, h = (blk / nb) % H, b = blk / (nb * H);
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const uint16_t* base = x + b * x_sb + (j * 128) * x_ss + h * x_sh + sub * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float f[8];
            unpack8<T>(*reinterpret_cast<const uint4*>(base + (long long)(tg * 8 + i) * x_ss), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += f[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) red[tg][sub * 8 + e] = acc[e];
        __syncthreads();
        if (threadIdx.x < 128) {
            float s = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) s += red[g][threadIdx.x];
            pooled[((b * H + h) * nb + j) * 128 + threadIdx.x] = from_f32<T>(s / 128.0f);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ V re-tiling
// vt[b][h][tile64][d = 0..127][pos = 0..63] with pos -> key = 32*(pos>>5) + pv_key_of_pos(pos&31): each lane
// of the attention kernel then reads its 8 P.V operand values of one d-row as ONE 16-byte LDS read.
// One workgroup per (b, h, tile64); transposition through LDS.
__global__ void pack_v_kernel(const uint16_t* __restrict__ v, uint16_t* __restrict__ vt, long long B, long long H,
                              long long ntile, long long v_sb, long long v_ss, long long v_sh, long long dst_tile0,
                              long long dst_ntile) {
    __shared__ uint16_t tile[64][136];  // +8 halfwords: rows stay 16-B aligned, column reads spread over banks
    for (long long wg = blockIdx.x; wg < B * H * ntile; wg += gridDim.x) {
        const long long tI = wg % ntile, h = (wg / ntile) % H, b = wg / (ntile * H);
        const uint16_t* src = v + b * v_sb + (tI * 64) * v_ss + h * v_sh;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = i * 16 + (threadIdx.x >> 4), c = threadIdx.x & 15;
            *reinterpret_cast<uint4*>(&tile[row][c * 8]) = ld_stream(src + (long long)row * v_ss + c * 8);
        }
        __syncthreads();
        uint16_t* dst = vt + (((b * H + h) * dst_ntile) + dst_tile0 + tI) * (128 * 64);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int chunk = i * 256 + threadIdx.x;   // 1024 chunks of 8 positions
            const int d = chunk >> 3, pc = chunk & 7;  // row d, positions pc*8 .. pc*8+7
            uint32_t w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int p0 = pc * 8 + 2 * e, p1 = p0 + 1;
                const int k0 = (p0 & 32) + pv_key_of_pos(p0 & 31), k1 = (p1 & 32) + pv_key_of_pos(p1 & 31);
                w[e] = (uint32_t)tile[k0][d] | ((uint32_t)tile[k1][d] << 16);
            }
            st_stream(dst + d * 64 + pc * 8, make_uint4(w[0], w[1], w[2], w[3]));
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ Ulysses pack
// x[b, s, r*Hn + hl, :]  <->  buf[r][b][s][hl][:]   (Hn = H / N); one 256-byte head row per 16 lanes.
template <bool PACK>
__global__ void ulysses_heads_kernel(const uint4* __restrict__ in, uint4* __restrict__ outp, long long B, long long S,
                                     long long H, long long N, long long sb, long long ss, long long sh) {
    const long long Hn = H / N, rows = B * S * H;
    const int sub = threadIdx.x & 15, rig = threadIdx.x >> 4;
    for (long long row = (long long)blockIdx.x * 16 + rig; row < rows; row += (long long)gridDim.x * 16) {
        const long long h = row % H, s = (row / H) % S, b = row / (H * S);
        const long long r = h / Hn, hl = h % Hn;
        const long long strided = (b * sb + s * ss + h * sh) / 8 + sub;                 // in uint4 units
        const long long packed = ((((r * B + b) * S + s) * Hn + hl) * 128) / 8 + sub;
        if (PACK)
            outp[packed] = in[strided];
        else
            outp[strided] = in[packed];
    }
}

inline int grid_for(long long work_items, int cap = 16384) {
    if (work_items < 1) work_items = 1;
    return (int)(work_items > cap ? cap : work_items);
}

// ---- Wan flavour of the block glue: the residual stream is fp32 (x + y*e under autocast(float32),
//      wan/modules/model_mul.py:331-343), the GEMM inputs are its LayerNorm rounded to the 16-bit dtype by autocast.
// y = cast( LN(x) [*w + b] [*(1 + scale) + shift] ); all arithmetic fp32 in the eager order (no contraction).
template <typename T>
__global__ void __launch_bounds__(256) wan_ln_modulate_kernel(const float* __restrict__ x, uint16_t* __restrict__ y,
                                                              const float* __restrict__ w,
                                                              const float* __restrict__ b,
                                                              const float* __restrict__ shift,
                                                              const float* __restrict__ scale, long long rows, int C,
                                                              long long x_rs, long long y_rs, float eps, int round_ln) {
    __shared__ float red[2][4];
    for (long long row = blockIdx.x; row < rows; row += gridDim.x) {
        const float* xr = x + row * x_rs;
        float f[6][4];   // up to 6144 channels: 256 threads x 4 floats x 6
        float s1 = 0.f;
#pragma unroll
        for (int nv = 0; nv < 6; ++nv) {
            const int c = threadIdx.x * 4 + nv * 1024;
            if (c >= C) break;
            const float4 v = *reinterpret_cast<const float4*>(xr + c);
            f[nv][0] = v.x; f[nv][1] = v.y; f[nv][2] = v.z; f[nv][3] = v.w;
            s1 += (v.x + v.y) + (v.z + v.w);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s1 += __shfl_xor(s1, o);
        if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = s1;
        __syncthreads();
        const float mean = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / (float)C;
        float s2 = 0.f;   // two-pass variance: the residual stream grows over 40 layers, E[x^2]-mean^2 would cancel
#pragma unroll
        for (int nv = 0; nv < 6; ++nv) {
            const int c = threadIdx.x * 4 + nv * 1024;
            if (c >= C) break;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = f[nv][e] - mean;
                s2 += d * d;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s2 += __shfl_xor(s2, o);
        if ((threadIdx.x & 63) == 0) red[1][threadIdx.x >> 6] = s2;
        __syncthreads();
        const float var = ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / (float)C;
        const float rstd = 1.0f / sqrtf(var + eps);
        __syncthreads();
#pragma unroll
        for (int nv = 0; nv < 6; ++nv) {
            const int c = threadIdx.x * 4 + nv * 1024;
            if (c >= C) break;
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = (f[nv][e] - mean) * rstd;
                if (w) t = t * w[c + e] + b[c + e];
                if (round_ln) t = round_to<T>(t);   // LayerNorm(...).type_as(x) of a 16-bit x (first block)
                if (scale) t = t * (1.0f + scale[c + e]) + shift[c + e];
                o[e] = t;
            }
            uint2 pk;
            pk.x = pack2<T>(o[0], o[1]);
            pk.y = pack2<T>(o[2], o[3]);
            *reinterpret_cast<uint2*>(y + row * y_rs + c) = pk;
        }
    }
}

// Wave-per-row form of wan_ln_modulate_kernel for widths C = 256 * NV (1536 -> NV 6, 5120 -> NV 20; round 3): a lane
// keeps its NV float4 of the row in registers, both statistics passes are wave shuffles, no barrier, four rows per
// workgroup in flight.  The block-per-row kernel above (two block-wide reductions per row) ran at 1.4 TB/s at C = 5120.
template <typename T, int NV, bool AFFINE, bool MOD>
__global__ void __launch_bounds__(256) wan_ln_modulate_wave_kernel(const float* __restrict__ x, uint16_t* __restrict__ y,
                                                                   const float* __restrict__ w,
                                                                   const float* __restrict__ b,
                                                                   const float* __restrict__ shift,
                                                                   const float* __restrict__ scale, long long rows,
                                                                   long long x_rs, long long y_rs, float eps,
                                                                   int round_ln) {
    constexpr int C = NV * 256;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long long row = (long long)blockIdx.x * 4 + wave; row < rows; row += (long long)gridDim.x * 4) {
        const float4* xr = reinterpret_cast<const float4*>(x + row * x_rs) + lane;
        float4 f[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) f[i] = xr[64 * i];
        float s1 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) s1 += (f[i].x + f[i].y) + (f[i].z + f[i].w);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s1 += __shfl_xor(s1, o);
        const float mean = s1 / (float)C;
        float s2 = 0.f;   // two-pass variance (see the block-per-row kernel)
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const float d0 = f[i].x - mean, d1 = f[i].y - mean, d2 = f[i].z - mean, d3 = f[i].w - mean;
            s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s2 += __shfl_xor(s2, o);
        const float rstd = 1.0f / sqrtf(s2 / (float)C + eps);
        uint16_t* yr = y + row * y_rs + lane * 4;
        // the parameter vectors depend on the column only: branch-free loads (template flags), four chunks in flight
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + 64 * i) * 4;
            float o[4] = {(f[i].x - mean) * rstd, (f[i].y - mean) * rstd, (f[i].z - mean) * rstd, (f[i].w - mean) * rstd};
            if (AFFINE) {
                const float4 wv = *reinterpret_cast<const float4*>(w + c), bv = *reinterpret_cast<const float4*>(b + c);
                o[0] = o[0] * wv.x + bv.x; o[1] = o[1] * wv.y + bv.y; o[2] = o[2] * wv.z + bv.z; o[3] = o[3] * wv.w + bv.w;
            }
            if (round_ln) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = round_to<T>(o[e]);
            }
            if (MOD) {
                const float4 sc = *reinterpret_cast<const float4*>(scale + c), sh = *reinterpret_cast<const float4*>(shift + c);
                o[0] = o[0] * (1.0f + sc.x) + sh.x; o[1] = o[1] * (1.0f + sc.y) + sh.y;
                o[2] = o[2] * (1.0f + sc.z) + sh.z; o[3] = o[3] * (1.0f + sc.w) + sh.w;
            }
            uint2 pk;
            pk.x = pack2<T>(o[0], o[1]);
            pk.y = pack2<T>(o[2], o[3]);
            *reinterpret_cast<uint2*>(yr + 256 * i) = pk;
            if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// out = x + float(y) [* gate]   (x, out fp32; y 16-bit; gate fp32 [C] or null)
template <typename T>
__global__ void wan_gate_residual_kernel(const float* x, const uint16_t* __restrict__ y,
                                         const float* __restrict__ gate, float* out, long long rows,
                                         int C, long long x_rs, long long y_rs, long long o_rs) {
    const int vpr = C / 8;
    const long long total = rows * vpr;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / vpr;
        const int c = (int)(i % vpr) * 8;
        float yv[8];
        unpack8<T>(*reinterpret_cast<const uint4*>(y + row * y_rs + c), yv);
        const float4 a0 = *reinterpret_cast<const float4*>(x + row * x_rs + c);
        const float4 a1 = *reinterpret_cast<const float4*>(x + row * x_rs + c + 4);
        float a[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        float gv[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
        if (gate) {
            const float4 g0 = *reinterpret_cast<const float4*>(gate + c), g1 = *reinterpret_cast<const float4*>(gate + c + 4);
            gv[0] = g0.x; gv[1] = g0.y; gv[2] = g0.z; gv[3] = g0.w; gv[4] = g1.x; gv[5] = g1.y; gv[6] = g1.z; gv[7] = g1.w;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = a[e] + (gate ? yv[e] * gv[e] : yv[e]);
        *reinterpret_cast<float4*>(out + row * o_rs + c) = make_float4(a[0], a[1], a[2], a[3]);
        *reinterpret_cast<float4*>(out + row * o_rs + c + 4) = make_float4(a[4], a[5], a[6], a[7]);
    }
}

// ------------------------------------------------------------------------------------------------ stream delay
// Measurement aid (bench.py --simulate-ranks): ONE wavefront that keeps its stream busy for a given time of the
// constant-rate wall clock (s_memrealtime) -- the stand-in for an xGMI transfer when a multi-rank job is replayed on a
// single GPU.  It occupies one wave slot, so the compute stream beside it keeps (almost) the whole chip, as it does
// beside an RCCL kernel that moves data with a handful of workgroups.
__global__ void __launch_bounds__(64) stream_delay_kernel(long long ticks) {
    const long long t0 = (long long)wall_clock64();
    while ((long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

}  // namespace
}  // namespace jenga

using namespace jenga;

#define JENGA_CHECK_LAUNCH(name)                          \
    do {                                                  \
        hipError_t e_ = hipGetLastError();                \
        if (e_ != hipSuccess) {                           \
            set_error(name ": %s", hipGetErrorString(e_)); \
            return JENGA_ELAUNCH;                         \
        }                                                 \
    } while (0)

extern "C" int jenga_gather_rows(void* stream, const void* src, void* dst, const int64_t* index, int64_t batch,
                                 int64_t n_rows, int64_t row_bytes, int64_t src_bs, int64_t dst_bs) {
    if (!src || !dst || !index || batch < 0 || n_rows < 0 || row_bytes <= 0 || (row_bytes & 15) || (src_bs & 15) ||
        (dst_bs & 15) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) {
        set_error("jenga_gather_rows: rows must be 16-byte multiples and 16-byte aligned (row_bytes=%lld)",
                  (long long)row_bytes);
        return JENGA_EINVAL;
    }
    if (batch * n_rows == 0) return JENGA_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(batch * n_rows, 65536)), dim3(256), 0, (hipStream_t)stream,
                       (const uint4*)src, (uint4*)dst, index, (long long)batch, (long long)n_rows,
                       (int)(row_bytes / 16), (long long)(src_bs / 16), (long long)(dst_bs / 16));
    JENGA_CHECK_LAUNCH("jenga_gather_rows");
    return JENGA_OK;
}

static bool strides_ok(int64_t a, int64_t b, int64_t c) { return !(a & 7) && !(b & 7) && !(c & 7); }

extern "C" int jenga_rmsnorm_rope(void* stream, const void* x, void* out, const void* weight, const float* cosT,
                                  const float* sinT, int64_t B, int64_t S, int64_t H, int64_t x_sb, int64_t x_ss,
                                  int64_t x_sh, int64_t o_sb, int64_t o_ss, int64_t o_sh, int64_t s_rope, float eps,
                                  int dtype) {
    if (!x || !out || B < 0 || S < 0 || H < 0 || !strides_ok(x_sb, x_ss, x_sh) || !strides_ok(o_sb, o_ss, o_sh) ||
        ((uintptr_t)x & 15) || ((uintptr_t)out & 15) || ((cosT == nullptr) != (sinT == nullptr))) {
        set_error("jenga_rmsnorm_rope: bad arguments (strides must be multiples of 8 elements, pointers 16-B aligned)");
        return JENGA_EINVAL;
    }
    if (dtype != JENGA_BF16 && dtype != JENGA_FP16) {
        set_error("jenga_rmsnorm_rope: dtype must be bf16 or fp16");
        return JENGA_EUNSUPPORTED;
    }
    const long long rows = (long long)B * S * H;
    if (rows == 0) return JENGA_OK;
    const int grid = grid_for((rows + 15) / 16, 32768);
#define LAUNCH_NR(T)                                                                                               \
    hipLaunchKernelGGL(rmsnorm_rope_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x,  \
                       (uint16_t*)out, (const uint16_t*)weight, cosT, sinT, (long long)B, (long long)S,            \
                       (long long)H, (long long)x_sb, (long long)x_ss, (long long)x_sh, (long long)o_sb,           \
                       (long long)o_ss, (long long)o_sh, (long long)s_rope, eps)
    if (dtype == JENGA_BF16) LAUNCH_NR(BF16); else LAUNCH_NR(FP16);
#undef LAUNCH_NR
    JENGA_CHECK_LAUNCH("jenga_rmsnorm_rope");
    return JENGA_OK;
}

extern "C" int jenga_qk_norm_rope_pool(void* stream, const void* xq, const void* xk, void* oq, void* ok,
                                       const void* wq, const void* wk, const float* cosT, const float* sinT,
                                       void* qpool, void* kpool, int64_t B, int64_t n_blocks, int64_t H, int64_t x_sb,
                                       int64_t x_ss, int64_t x_sh, int64_t o_sb, int64_t o_ss, int64_t o_sh,
                                       int64_t s_rope, int64_t pool_block0, int64_t nq_pool, int64_t nk_pool, float eps,
                                       int dtype) {
    if (!xq || !xk || !oq || !ok || B < 0 || n_blocks < 0 || H <= 0 || H > 64 || !strides_ok(x_sb, x_ss, x_sh) ||
        !strides_ok(o_sb, o_ss, o_sh) || ((uintptr_t)xq & 15) || ((uintptr_t)xk & 15) || ((uintptr_t)oq & 15) ||
        ((uintptr_t)ok & 15) || ((cosT == nullptr) != (sinT == nullptr)) || pool_block0 < 0) {
        set_error("jenga_qk_norm_rope_pool: bad arguments (1 <= H <= 64, strides multiples of 8 elements, pointers "
                  "16-B aligned)");
        return JENGA_EINVAL;
    }
    if (dtype != JENGA_BF16 && dtype != JENGA_FP16) {
        set_error("jenga_qk_norm_rope_pool: dtype must be bf16 or fp16");
        return JENGA_EUNSUPPORTED;
    }
    const long long n = (long long)B * n_blocks;
    if (n == 0) return JENGA_OK;
    int HG = 1;   // heads per workgroup: the largest divisor of H that is <= 8 (two waves: fine-grained, full occupancy)
    for (int d = 8; d >= 1; --d)
        if (H % d == 0) { HG = d; break; }
    const long long n_wg = (n + 7) / 8 * 8 * (H / HG);
#define LAUNCH_QKP(T)                                                                                             \
    hipLaunchKernelGGL(qk_norm_rope_pool_kernel<T>, dim3(grid_for(n_wg, 1 << 20)), dim3((unsigned)(16 * HG)), 0,  \
                       (hipStream_t)stream, (const uint16_t*)xq, (const uint16_t*)xk, (uint16_t*)oq, (uint16_t*)ok, \
                       (const uint16_t*)wq, (const uint16_t*)wk, cosT, sinT, (uint16_t*)qpool, (uint16_t*)kpool,  \
                       (long long)B, (long long)n_blocks, (long long)H, (long long)x_sb, (long long)x_ss,          \
                       (long long)x_sh, (long long)o_sb, (long long)o_ss, (long long)o_sh, (long long)s_rope,     \
                       (long long)pool_block0, (long long)nq_pool, (long long)nk_pool, eps, HG)
    if (dtype == JENGA_BF16) LAUNCH_QKP(BF16); else LAUNCH_QKP(FP16);
#undef LAUNCH_QKP
    JENGA_CHECK_LAUNCH("jenga_qk_norm_rope_pool");
    return JENGA_OK;
}

extern "C" int jenga_sp_qkv_prologue(void* stream, const void* xq, const void* xk, const void* xv, void* oq, void* ok,
                                     void* ov, const void* wq, const void* wk, const float* cosT, const float* sinT,
                                     int64_t B, int64_t S, int64_t H, int64_t head0, int64_t n_heads,
                                     int64_t heads_per_peer, int64_t x_sb, int64_t x_ss, int64_t x_sh, int64_t o_sp,
                                     int64_t o_sb, int64_t o_ss, int64_t o_sh, int64_t s_rope, float eps, int dtype) {
    const bool has_qk = xq != nullptr, has_v = xv != nullptr;
    if ((!has_qk && !has_v) || (has_qk && (!xk || !oq || !ok)) || (!has_qk && xk) || (has_v && !ov) || B < 0 || S < 0 ||
        H <= 0 || head0 < 0 || n_heads < 0 ||
        head0 + n_heads > H || heads_per_peer <= 0 || !strides_ok(x_sb, x_ss, x_sh) || !strides_ok(o_sb, o_ss, o_sh) ||
        (o_sp & 7) || o_sp < 0 || ((uintptr_t)xq & 15) || ((uintptr_t)xk & 15) || ((uintptr_t)xv & 15) ||
        ((uintptr_t)oq & 15) || ((uintptr_t)ok & 15) || ((uintptr_t)ov & 15) ||
        ((cosT == nullptr) != (sinT == nullptr)) || s_rope < 0) {
        set_error("jenga_sp_qkv_prologue: bad arguments (head window [%lld, %lld) of %lld heads, %lld heads per peer; "
                  "strides multiples of 8 elements, pointers 16-B aligned)",
                  (long long)head0, (long long)(head0 + n_heads), (long long)H, (long long)heads_per_peer);
        return JENGA_EINVAL;
    }
    if (dtype != JENGA_BF16 && dtype != JENGA_FP16) {
        set_error("jenga_sp_qkv_prologue: dtype must be bf16 or fp16");
        return JENGA_EUNSUPPORTED;
    }
    const long long rows = (long long)B * S * n_heads;
    if (rows == 0) return JENGA_OK;
    const int grid = grid_for((rows + 15) / 16, 65536);
#define LAUNCH_SPP(T)                                                                                              \
    hipLaunchKernelGGL(sp_qkv_prologue_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream,                   \
                       (const uint16_t*)xq, (const uint16_t*)xk, (const uint16_t*)xv, (uint16_t*)oq, (uint16_t*)ok, \
                       (uint16_t*)ov, (const uint16_t*)wq, (const uint16_t*)wk, cosT, sinT, (long long)B,          \
                       (long long)S, (long long)head0, (long long)n_heads, (long long)heads_per_peer,              \
                       (long long)x_sb, (long long)x_ss, (long long)x_sh, (long long)o_sp, (long long)o_sb,        \
                       (long long)o_ss, (long long)o_sh, (long long)s_rope, eps)
    if (dtype == JENGA_BF16) LAUNCH_SPP(BF16); else LAUNCH_SPP(FP16);
#undef LAUNCH_SPP
    JENGA_CHECK_LAUNCH("jenga_sp_qkv_prologue");
    return JENGA_OK;
}

extern "C" int jenga_block_pool(void* stream, const void* x, void* pooled, int64_t B, int64_t H, int64_t n_blocks,
                                int64_t x_sb, int64_t x_ss, int64_t x_sh, int dtype) {
    if (!x || !pooled || B < 0 || H < 0 || n_blocks < 0 || !strides_ok(x_sb, x_ss, x_sh) || ((uintptr_t)x & 15)) {
        set_error("jenga_block_pool: bad arguments");
        return JENGA_EINVAL;
    }
    if (dtype != JENGA_BF16 && dtype != JENGA_FP16) {
        set_error("jenga_block_pool: dtype must be bf16 or fp16");
        return JENGA_EUNSUPPORTED;
    }
    const long long n = (long long)B * H * n_blocks;
    if (n == 0) return JENGA_OK;
#define LAUNCH_BP(T)                                                                                              \
    hipLaunchKernelGGL(block_pool_kernel<T>, dim3(grid_for(n, 65536)), dim3(256), 0, (hipStream_t)stream,         \
                       (const uint16_t*)x, (uint16_t*)pooled, (long long)B, (long long)H, (long long)n_blocks,    \
                       (long long)x_sb, (long long)x_ss, (long long)x_sh)
    if (dtype == JENGA_BF16) LAUNCH_BP(BF16); else LAUNCH_BP(FP16);
#undef LAUNCH_BP
    JENGA_CHECK_LAUNCH("jenga_block_pool");
    return JENGA_OK;
}

extern "C" size_t jenga_pack_v_bytes(int64_t B, int64_t H, int64_t n_blocks) {
    return (size_t)B * H * n_blocks * 128 * 128 * 2;
}

extern "C" int jenga_pack_v(void* stream, const void* v, void* vt, int64_t B, int64_t H, int64_t n_blocks,
                            int64_t v_sb, int64_t v_ss, int64_t v_sh, int64_t dst_block0, int64_t dst_blocks_total,
                            int dtype) {
    if (!v || !vt || B < 0 || H < 0 || n_blocks < 0 || !strides_ok(v_sb, v_ss, v_sh) || ((uintptr_t)v & 15) ||
        ((uintptr_t)vt & 15) || dst_block0 < 0 || dst_block0 + n_blocks > dst_blocks_total) {
        set_error("jenga_pack_v: bad arguments");
        return JENGA_EINVAL;
    }
    (void)dtype;  // pure 16-bit data movement
    const long long n = (long long)B * H * n_blocks * 2;
    if (n == 0) return JENGA_OK;
    hipLaunchKernelGGL(pack_v_kernel, dim3(grid_for(n, 65536)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)v,
                       (uint16_t*)vt, (long long)B, (long long)H, (long long)(n_blocks * 2), (long long)v_sb,
                       (long long)v_ss, (long long)v_sh, (long long)(dst_block0 * 2), (long long)(dst_blocks_total * 2));
    JENGA_CHECK_LAUNCH("jenga_pack_v");
    return JENGA_OK;
}

static int ulysses_common(bool pack, void* stream, const void* in, void* outp, int64_t B, int64_t S, int64_t H,
                          int64_t N, int64_t sb, int64_t ss, int64_t sh) {
    if (!in || !outp || N <= 0 || H % N || !strides_ok(sb, ss, sh) || ((uintptr_t)in & 15) || ((uintptr_t)outp & 15)) {
        set_error("jenga_ulysses_%s_heads: bad arguments (H=%lld must be divisible by N=%lld)", pack ? "pack" : "unpack",
                  (long long)H, (long long)N);
        return JENGA_EINVAL;
    }
    const long long rows = (long long)B * S * H;
    if (rows == 0) return JENGA_OK;
    const int grid = grid_for((rows + 15) / 16, 32768);
    if (pack)
        hipLaunchKernelGGL(ulysses_heads_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint4*)in,
                           (uint4*)outp, (long long)B, (long long)S, (long long)H, (long long)N, (long long)sb,
                           (long long)ss, (long long)sh);
    else
        hipLaunchKernelGGL(ulysses_heads_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           (const uint4*)in, (uint4*)outp, (long long)B, (long long)S, (long long)H, (long long)N,
                           (long long)sb, (long long)ss, (long long)sh);
    JENGA_CHECK_LAUNCH("jenga_ulysses_heads");
    return JENGA_OK;
}

extern "C" int jenga_ulysses_pack_heads(void* stream, const void* x, void* send, int64_t B, int64_t S_loc, int64_t H,
                                        int64_t N, int64_t x_sb, int64_t x_ss, int64_t x_sh) {
    return ulysses_common(true, stream, x, send, B, S_loc, H, N, x_sb, x_ss, x_sh);
}
extern "C" int jenga_ulysses_unpack_heads(void* stream, const void* recv, void* y, int64_t B, int64_t S_loc, int64_t H,
                                          int64_t N, int64_t y_sb, int64_t y_ss, int64_t y_sh) {
    return ulysses_common(false, stream, recv, y, B, S_loc, H, N, y_sb, y_ss, y_sh);
}

// ================================================================================================ Wan flavour
// Full-width RMSNorm (wan/modules/model_mul.py:74-90): y = (x.float() * rsqrt(mean(x^2) + eps)).type_as(x) * weight
// with weight in fp32 (out fp32, torch type promotion) or in the storage dtype (out in the storage dtype).
// One workgroup per row; C <= 8192, multiple of 8.
namespace jenga {
namespace {
template <typename T, bool W32>
__global__ void __launch_bounds__(256) rmsnorm_rows_kernel(const uint16_t* __restrict__ x, void* __restrict__ out,
                                                           const void* __restrict__ weight, long long rows, int C,
                                                           long long x_rs, long long o_rs, float eps) {
    __shared__ float red[4];
    for (long long row = blockIdx.x; row < rows; row += gridDim.x) {
        const uint16_t* xr = x + row * x_rs;
        float f[4][8];
        float ss = 0.f;
        int nv = 0;
        for (int c = threadIdx.x * 8; c < C; c += 256 * 8, ++nv) {
            unpack8<T>(*reinterpret_cast<const uint4*>(xr + c), f[nv]);
#pragma unroll
            for (int e = 0; e < 8; ++e) ss = __fadd_rn(ss, __fmul_rn(f[nv][e], f[nv][e]));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
        __syncthreads();
        ss = (red[0] + red[1]) + (red[2] + red[3]);
        __syncthreads();
        const float r = __fdiv_rn(1.0f, __fsqrt_rn(__fadd_rn(__fdiv_rn(ss, (float)C), eps)));
        nv = 0;
        for (int c = threadIdx.x * 8; c < C; c += 256 * 8, ++nv) {
            float y[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = round_to<T>(__fmul_rn(f[nv][e], r));
            if (W32) {
                const float4* wp = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(weight) + c);
                const float4 w0 = wp[0], w1 = wp[1];
                float4* op = reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + row * o_rs + c);
                op[0] = make_float4(__fmul_rn(y[0], w0.x), __fmul_rn(y[1], w0.y), __fmul_rn(y[2], w0.z),
                                    __fmul_rn(y[3], w0.w));
                op[1] = make_float4(__fmul_rn(y[4], w1.x), __fmul_rn(y[5], w1.y), __fmul_rn(y[6], w1.z),
                                    __fmul_rn(y[7], w1.w));
            } else {
                float wv[8];
                unpack8<T>(*reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(weight) + c), wv);
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = __fmul_rn(y[e], wv[e]);
                *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(out) + row * o_rs + c) = pack8<T>(y);
            }
        }
    }
}

// Wan self-attention prologue in one pass (wan/modules/model_mul.py:146-151 + the bf16 cast of the attention op,
// wan/modules/attention_block_triton_diffres.py:456-463): WanRMSNorm over the full model width (fp32 weight, so the
// reference's product is fp32), then the float64 complex RoPE on tokens < s_rope, then ONE cast to bf16 -- exactly the
// arithmetic of rmsnorm_rows_kernel<T, true> followed by rope_complex_kernel<2, 0>, without the two fp32 tensors in
// between (4.6 GB per tensor and layer at the Wan2.1-14B 720p shape).  cosT == nullptr: norm + cast only (the
// cross-attention's q).  One workgroup per token row; C % 8 == 0, C <= 8192; head_dim 128 (the table has 64 columns).
template <typename T>
__global__ void __launch_bounds__(256) wan_norm_rope_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ out,
                                                            const float* __restrict__ weight,
                                                            const double* __restrict__ cosT,
                                                            const double* __restrict__ sinT, long long rows, int C,
                                                            long long x_rs, long long o_rs, long long s_rope, float eps) {
    __shared__ float red[4];
    for (long long row = blockIdx.x; row < rows; row += gridDim.x) {
        const uint16_t* xr = x + row * x_rs;
        float f[4][8];
        float ss = 0.f;
        int nv = 0;
        for (int c = threadIdx.x * 8; c < C; c += 256 * 8, ++nv) {
            unpack8<T>(*reinterpret_cast<const uint4*>(xr + c), f[nv]);
#pragma unroll
            for (int e = 0; e < 8; ++e) ss = __fadd_rn(ss, __fmul_rn(f[nv][e], f[nv][e]));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
        __syncthreads();
        ss = (red[0] + red[1]) + (red[2] + red[3]);
        __syncthreads();
        const float r = __fdiv_rn(1.0f, __fsqrt_rn(__fadd_rn(__fdiv_rn(ss, (float)C), eps)));
        const bool rope = cosT && row < s_rope;
        nv = 0;
        for (int c = threadIdx.x * 8; c < C; c += 256 * 8, ++nv) {
            const float4* wp = reinterpret_cast<const float4*>(weight + c);
            const float4 w0 = wp[0], w1 = wp[1];
            const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
            float y[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = __fmul_rn(round_to<T>(__fmul_rn(f[nv][e], r)), wv[e]);   // fp32 product
            if (rope) {
                const double* cp = cosT + row * 64 + ((c & 127) >> 1);
                const double* sp = sinT + row * 64 + ((c & 127) >> 1);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const double a = (double)y[2 * j], bb = (double)y[2 * j + 1], cc = cp[j], d = sp[j];
                    const double re = __dsub_rn(__dmul_rn(a, cc), __dmul_rn(bb, d));
                    const double im = __dadd_rn(__dmul_rn(a, d), __dmul_rn(bb, cc));
                    y[2 * j] = (float)re;
                    y[2 * j + 1] = (float)im;
                }
            }
            *reinterpret_cast<uint4*>(out + row * o_rs + c) = pack8<BF16>(y);
        }
    }
}

// fp64 complex RoPE (wan/modules/model_mul.py:40-71): pairs (x[2j], x[2j+1]) as complex, multiplied in float64 by
// (cos, sin)[s][j] (tables already expanded per position, incl. the Hilbert freq_remap), rounded to fp32 (".float()")
// or further to the storage dtype when the consumer is the bf16 attention op.  Tokens s >= s_rope pass through.
// IN: 0 = bf16, 1 = fp16, 2 = fp32; OUT: 0 = bf16, 2 = fp32
template <int IN, int OUT>
__global__ void rope_complex_kernel(const void* __restrict__ x, void* __restrict__ out, const double* __restrict__ cosT,
                                    const double* __restrict__ sinT, long long B, long long S, long long H,
                                    long long x_sb, long long x_ss, long long x_sh, long long o_sb, long long o_ss,
                                    long long o_sh, long long s_rope) {
    const int sub = threadIdx.x & 15, rig = threadIdx.x >> 4;
    const long long rows = B * S * H;
    for (long long row = (long long)blockIdx.x * 16 + rig; row < rows; row += (long long)gridDim.x * 16) {
        const long long h = row % H, s = (row / H) % S, b = row / (H * S);
        const long long xo = b * x_sb + s * x_ss + h * x_sh + sub * 8;
        float f[8];
        if (IN == 2) {
            const float4* p = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(x) + xo);
            const float4 a = p[0], c = p[1];
            f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = c.x; f[5] = c.y; f[6] = c.z; f[7] = c.w;
        } else if (IN == 0) {
            unpack8<BF16>(*reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(x) + xo), f);
        } else {
            unpack8<FP16>(*reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(x) + xo), f);
        }
        if (s < s_rope) {
            const double* cp = cosT + s * 64 + sub * 4;
            const double* sp = sinT + s * 64 + sub * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double a = (double)f[2 * j], bb = (double)f[2 * j + 1], c = cp[j], d = sp[j];
                const double re = __dsub_rn(__dmul_rn(a, c), __dmul_rn(bb, d));   // (a+ib)(c+id), no fused ops
                const double im = __dadd_rn(__dmul_rn(a, d), __dmul_rn(bb, c));
                f[2 * j] = (float)re;
                f[2 * j + 1] = (float)im;
            }
        }
        const long long oo = b * o_sb + s * o_ss + h * o_sh + sub * 8;
        if (OUT == 2) {
            float4* p = reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + oo);
            p[0] = make_float4(f[0], f[1], f[2], f[3]);
            p[1] = make_float4(f[4], f[5], f[6], f[7]);
        } else {
            *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(out) + oo) = pack8<BF16>(f);
        }
    }
}
}  // namespace
}  // namespace jenga

extern "C" int jenga_rmsnorm_rows(void* stream, const void* x, void* out, const void* weight, int64_t rows, int64_t C,
                                  int64_t x_row_stride, int64_t o_row_stride, float eps, int dtype, int weight_fp32) {
    if (!x || !out || !weight || rows < 0 || C <= 0 || (C & 7) || C > 8192 || (x_row_stride & 7) ||
        (o_row_stride & 7) || ((uintptr_t)x & 15) || ((uintptr_t)out & 15) || ((uintptr_t)weight & 15)) {
        set_error("jenga_rmsnorm_rows: bad arguments (C must be a multiple of 8, <= 8192)");
        return JENGA_EINVAL;
    }
    if (dtype != JENGA_BF16 && dtype != JENGA_FP16) {
        set_error("jenga_rmsnorm_rows: dtype must be bf16 or fp16");
        return JENGA_EUNSUPPORTED;
    }
    if (rows == 0) return JENGA_OK;
    const int grid = grid_for(rows, 65536);
#define LAUNCH_RR(T, W)                                                                                          \
    hipLaunchKernelGGL((rmsnorm_rows_kernel<T, W>), dim3(grid), dim3(256), 0, (hipStream_t)stream,               \
                       (const uint16_t*)x, out, weight, (long long)rows, (int)C, (long long)x_row_stride,        \
                       (long long)o_row_stride, eps)
    if (dtype == JENGA_BF16) { if (weight_fp32) LAUNCH_RR(BF16, true); else LAUNCH_RR(BF16, false); }
    else { if (weight_fp32) LAUNCH_RR(FP16, true); else LAUNCH_RR(FP16, false); }
#undef LAUNCH_RR
    JENGA_CHECK_LAUNCH("jenga_rmsnorm_rows");
    return JENGA_OK;
}

extern "C" int jenga_wan_norm_rope(void* stream, const void* x, void* out, const float* weight, const double* cosT,
                                   const double* sinT, int64_t rows, int64_t C, int64_t x_row_stride,
                                   int64_t o_row_stride, int64_t s_rope, float eps, int dtype) {
    if (!x || !out || !weight || rows < 0 || C <= 0 || (C & 7) || C > 8192 || (x_row_stride & 7) || (o_row_stride & 7) ||
        ((uintptr_t)x & 15) || ((uintptr_t)out & 15) || ((uintptr_t)weight & 15) || ((cosT == nullptr) != (sinT == nullptr)) ||
        s_rope < 0 || (cosT && (C & 127)) || (cosT && s_rope > rows)) {
        set_error("jenga_wan_norm_rope: bad arguments (C multiple of 8 -- of 128 with RoPE --, <= 8192; the rows are ONE "
                  "sequence whose row index is the RoPE position: s_rope <= rows)");
        return JENGA_EINVAL;
    }
    if (dtype != JENGA_BF16 && dtype != JENGA_FP16) {
        set_error("jenga_wan_norm_rope: dtype must be bf16 or fp16");
        return JENGA_EUNSUPPORTED;
    }
    if (rows == 0) return JENGA_OK;
    const int grid = grid_for(rows, 65536);
    if (dtype == JENGA_BF16)
        hipLaunchKernelGGL(wan_norm_rope_kernel<BF16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x,
                           (uint16_t*)out, weight, cosT, sinT, (long long)rows, (int)C, (long long)x_row_stride,
                           (long long)o_row_stride, (long long)s_rope, eps);
    else
        hipLaunchKernelGGL(wan_norm_rope_kernel<FP16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x,
                           (uint16_t*)out, weight, cosT, sinT, (long long)rows, (int)C, (long long)x_row_stride,
                           (long long)o_row_stride, (long long)s_rope, eps);
    JENGA_CHECK_LAUNCH("jenga_wan_norm_rope");
    return JENGA_OK;
}

extern "C" int jenga_rope_complex(void* stream, const void* x, void* out, const double* cosT, const double* sinT,
                                  int64_t B, int64_t S, int64_t H, int64_t x_sb, int64_t x_ss, int64_t x_sh,
                                  int64_t o_sb, int64_t o_ss, int64_t o_sh, int64_t s_rope, int in_dtype,
                                  int out_dtype) {
    if (!x || !out || !cosT || !sinT || B < 0 || S < 0 || H < 0 || !strides_ok(x_sb, x_ss, x_sh) ||
        !strides_ok(o_sb, o_ss, o_sh) || s_rope < 0 || s_rope > S) {
        set_error("jenga_rope_complex: bad arguments");
        return JENGA_EINVAL;
    }
    if ((in_dtype != 0 && in_dtype != 1 && in_dtype != 2) || (out_dtype != 0 && out_dtype != 2)) {
        set_error("jenga_rope_complex: in_dtype in {bf16=0, fp16=1, fp32=2}, out_dtype in {bf16=0, fp32=2}");
        return JENGA_EUNSUPPORTED;
    }
    const long long rows = (long long)B * S * H;
    if (rows == 0) return JENGA_OK;
    const int grid = grid_for((rows + 15) / 16, 32768);
#define LAUNCH_RC(I, O)                                                                                           \
    hipLaunchKernelGGL((rope_complex_kernel<I, O>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x, out, cosT,  \
                       sinT, (long long)B, (long long)S, (long long)H, (long long)x_sb, (long long)x_ss,          \
                       (long long)x_sh, (long long)o_sb, (long long)o_ss, (long long)o_sh, (long long)s_rope)
    if (out_dtype == 2) { if (in_dtype == 0) LAUNCH_RC(0, 2); else if (in_dtype == 1) LAUNCH_RC(1, 2); else LAUNCH_RC(2, 2); }
    else { if (in_dtype == 0) LAUNCH_RC(0, 0); else if (in_dtype == 1) LAUNCH_RC(1, 0); else LAUNCH_RC(2, 0); }
#undef LAUNCH_RC
    JENGA_CHECK_LAUNCH("jenga_rope_complex");
    return JENGA_OK;
}

// ================================================================================================ DiT block glue (f-2/f-3)
// Fused elementwise passes around the GEMMs of the DiT blocks (models_mul_block_gc_ha_multigpu.py:196-203, 297-315,
// 404-406, 498-500; hyvideo_i2v/modules/modulate_layers.py:37-110 for the token-replace variant).  Rounding points
// follow the eager reference: every torch op result is rounded to the storage dtype.
namespace jenga {
namespace {

// y = modulate(LayerNorm(x), shift, scale): LayerNorm without affine (eps), then x*(1+scale)+shift.  With a token
// mask (I2V "token_replace": first-frame tokens use a second modulation set) rows with mask != 0 use (shift2, scale2).
// One workgroup per row, C <= 8192 (multiple of 8).
template <typename T>
__global__ void __launch_bounds__(256) ln_modulate_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y,
                                                          const uint16_t* __restrict__ shift,
                                                          const uint16_t* __restrict__ scale,
                                                          const uint16_t* __restrict__ shift2,
                                                          const uint16_t* __restrict__ scale2,
                                                          const uint8_t* __restrict__ mask, long long rows, int C,
                                                          long long x_rs, long long y_rs, float eps) {
    __shared__ float red[2][4];
    for (long long row = blockIdx.x; row < rows; row += gridDim.x) {
        const uint16_t* xr = x + row * x_rs;
        const bool alt = mask && mask[row];
        const uint16_t* sh = alt ? shift2 : shift;
        const uint16_t* sc = alt ? scale2 : scale;
        float f[4][8];
        float s1 = 0.f, s2 = 0.f;
        int nv = 0;
        for (int c = threadIdx.x * 8; c < C; c += 2048, ++nv) {
            unpack8<T>(*reinterpret_cast<const uint4*>(xr + c), f[nv]);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s1 += f[nv][e];
                s2 += f[nv][e] * f[nv][e];
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            s1 += __shfl_xor(s1, o);
            s2 += __shfl_xor(s2, o);
        }
        if ((threadIdx.x & 63) == 0) {
            red[0][threadIdx.x >> 6] = s1;
            red[1][threadIdx.x >> 6] = s2;
        }
        __syncthreads();
        s1 = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        s2 = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        __syncthreads();
        const float mean = s1 / (float)C;
        const float var = fmaxf(s2 / (float)C - mean * mean, 0.f);
        const float rstd = 1.0f / sqrtf(var + eps);
        nv = 0;
        for (int c = threadIdx.x * 8; c < C; c += 2048, ++nv) {
            float sv[8], hv[8], o[8];
            unpack8<T>(*reinterpret_cast<const uint4*>(sc + c), sv);
            unpack8<T>(*reinterpret_cast<const uint4*>(sh + c), hv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float n = round_to<T>((f[nv][e] - mean) * rstd);           // F.layer_norm -> dtype
                const float m = round_to<T>(n * round_to<T>(1.0f + sv[e]));       // x * (1 + scale)
                o[e] = m + hv[e];                                                 // + shift (rounded by pack8)
            }
            *reinterpret_cast<uint4*>(y + row * y_rs + c) = pack8<T>(o);
        }
    }
}

// Wave-per-row form for C = 512 * NV (hidden 3072 -> NV 6; round 4): a lane keeps its NV x 8 values of the row in
// registers, both statistics are wave shuffles, no barrier, four rows per workgroup in flight -- the block-per-row kernel
// above (one block-wide reduction and two barriers per 6 KB row) ran at 3.7 TB/s at the 720p shape.  Same arithmetic
// and rounding points; the fp32 summation ORDER of the two row statistics differs from the block kernel's.
template <typename T, int NV>
__global__ void __launch_bounds__(256) ln_modulate_wave_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y,
                                                               const uint16_t* __restrict__ shift,
                                                               const uint16_t* __restrict__ scale,
                                                               const uint16_t* __restrict__ shift2,
                                                               const uint16_t* __restrict__ scale2,
                                                               const uint8_t* __restrict__ mask, long long rows,
                                                               long long x_rs, long long y_rs, float eps) {
    constexpr int C = NV * 512;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long long row = (long long)blockIdx.x * 4 + wave; row < rows; row += (long long)gridDim.x * 4) {
        const uint16_t* xr = x + row * x_rs + lane * 8;
        uint4 raw[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) raw[i] = ld_stream(xr + i * 512);
        const bool alt = mask && mask[row];
        const uint16_t* sh = (alt ? shift2 : shift) + lane * 8;
        const uint16_t* sc = (alt ? scale2 : scale) + lane * 8;
        float f[NV][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            unpack8<T>(raw[i], f[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s1 += f[i][e];
                s2 += f[i][e] * f[i][e];
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            s1 += __shfl_xor(s1, o);
            s2 += __shfl_xor(s2, o);
        }
        const float mean = s1 / (float)C;
        const float var = fmaxf(s2 / (float)C - mean * mean, 0.f);
        const float rstd = 1.0f / sqrtf(var + eps);
        uint16_t* yr = y + row * y_rs + lane * 8;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            float sv[8], hv[8], o[8];
            unpack8<T>(*reinterpret_cast<const uint4*>(sc + i * 512), sv);
            unpack8<T>(*reinterpret_cast<const uint4*>(sh + i * 512), hv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float n = round_to<T>((f[i][e] - mean) * rstd);            // F.layer_norm -> dtype
                const float m = round_to<T>(n * round_to<T>(1.0f + sv[e]));       // x * (1 + scale)
                o[e] = m + hv[e];                                                 // + shift (rounded by pack8)
            }
            st_stream(yr + i * 512, pack8<T>(o));
        }
    }
}

// out = res + y * gate (apply_gate + residual add); rows with mask != 0 use gate2.  16 B per lane, grid-stride.
template <typename T>
__global__ void gate_residual_kernel(const uint16_t* __restrict__ res, const uint16_t* __restrict__ y,
                                     const uint16_t* __restrict__ gate, const uint16_t* __restrict__ gate2,
                                     const uint8_t* __restrict__ mask, uint16_t* __restrict__ out, long long rows,
                                     int C, long long res_rs, long long y_rs, long long o_rs) {
    const int vpr = C / 8;
    const long long total = rows * vpr;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / vpr;
        const int c = (int)(i % vpr) * 8;
        const uint16_t* g = (mask && mask[row]) ? gate2 : gate;
        float a[8], b[8], gv[8], o[8];
        unpack8<T>(*reinterpret_cast<const uint4*>(res + row * res_rs + c), a);
        unpack8<T>(*reinterpret_cast<const uint4*>(y + row * y_rs + c), b);
        unpack8<T>(*reinterpret_cast<const uint4*>(g + c), gv);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = a[e] + round_to<T>(b[e] * gv[e]);
        *reinterpret_cast<uint4*>(out + row * o_rs + c) = pack8<T>(o);
    }
}

// tanh-approximated GELU from a (strided) source into a (strided) destination: the single-stream blocks write
// gelu(mlp) straight into the right part of linear2's concat buffer.
template <typename T>
__global__ void gelu_tanh_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ out, long long rows, int C,
                                 long long x_rs, long long o_rs) {
    const int vpr = C / 8;
    const long long total = rows * vpr;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / vpr;
        const int c = (int)(i % vpr) * 8;
        float a[8], o[8];
        unpack8<T>(*reinterpret_cast<const uint4*>(x + row * x_rs + c), a);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = a[e];
            const float inner = 0.7978845608028654f * (v + 0.044715f * v * v * v);
            o[e] = 0.5f * v * (1.0f + tanhf(inner));
        }
        *reinterpret_cast<uint4*>(out + row * o_rs + c) = pack8<T>(o);
    }
}

}  // namespace
}  // namespace jenga

extern "C" int jenga_ln_modulate(void* stream, const void* x, void* y, const void* shift, const void* scale,
                                 const void* shift2, const void* scale2, const uint8_t* mask, int64_t rows, int64_t C,
                                 int64_t x_row_stride, int64_t y_row_stride, float eps, int dtype) {
    if (!x || !y || !shift || !scale || rows < 0 || C <= 0 || (C & 7) || C > 8192 || (x_row_stride & 7) ||
        (y_row_stride & 7) || (mask && (!shift2 || !scale2))) {
        set_error("jenga_ln_modulate: bad arguments (C multiple of 8, <= 8192; mask needs the second modulation set)");
        return JENGA_EINVAL;
    }
    if (dtype != JENGA_BF16 && dtype != JENGA_FP16) {
        set_error("jenga_ln_modulate: dtype must be bf16 or fp16");
        return JENGA_EUNSUPPORTED;
    }
    if (rows == 0) return JENGA_OK;
#define LAUNCH_LM(T)                                                                                              \
    hipLaunchKernelGGL(ln_modulate_kernel<T>, dim3(grid_for(rows, 65536)), dim3(256), 0, (hipStream_t)stream,     \
                       (const uint16_t*)x, (uint16_t*)y, (const uint16_t*)shift, (const uint16_t*)scale,          \
                       (const uint16_t*)shift2, (const uint16_t*)scale2, mask, (long long)rows, (int)C,           \
                       (long long)x_row_stride, (long long)y_row_stride, eps)
#define LAUNCH_LMW(T, NV)                                                                                         \
    hipLaunchKernelGGL((ln_modulate_wave_kernel<T, NV>), dim3(grid_for((rows + 3) / 4, 65536)), dim3(256), 0,     \
                       (hipStream_t)stream, (const uint16_t*)x, (uint16_t*)y, (const uint16_t*)shift,             \
                       (const uint16_t*)scale, (const uint16_t*)shift2, (const uint16_t*)scale2, mask,            \
                       (long long)rows, (long long)x_row_stride, (long long)y_row_stride, eps)
    const bool aligned = !((uintptr_t)x & 15) && !((uintptr_t)y & 15) && !((uintptr_t)shift & 15) &&
                         !((uintptr_t)scale & 15) && !((uintptr_t)shift2 & 15) && !((uintptr_t)scale2 & 15);
    if (C == 3072 && aligned) {
        if (dtype == JENGA_BF16) LAUNCH_LMW(BF16, 6); else LAUNCH_LMW(FP16, 6);
    } else if (C == 1024 && aligned) {
        if (dtype == JENGA_BF16) LAUNCH_LMW(BF16, 2); else LAUNCH_LMW(FP16, 2);
    } else {
        if (dtype == JENGA_BF16) LAUNCH_LM(BF16); else LAUNCH_LM(FP16);
    }
#undef LAUNCH_LMW
#undef LAUNCH_LM
    JENGA_CHECK_LAUNCH("jenga_ln_modulate");
    return JENGA_OK;
}

extern "C" int jenga_gate_residual(void* stream, const void* res, const void* y, const void* gate, const void* gate2,
                                   const uint8_t* mask, void* out, int64_t rows, int64_t C, int64_t res_row_stride,
                                   int64_t y_row_stride, int64_t o_row_stride, int dtype) {
    if (!res || !y || !gate || !out || rows < 0 || C <= 0 || (C & 7) || (res_row_stride & 7) || (y_row_stride & 7) ||
        (o_row_stride & 7) || (mask && !gate2)) {
        set_error("jenga_gate_residual: bad arguments");
        return JENGA_EINVAL;
    }
    if (dtype != JENGA_BF16 && dtype != JENGA_FP16) {
        set_error("jenga_gate_residual: dtype must be bf16 or fp16");
        return JENGA_EUNSUPPORTED;
    }
    if (rows == 0) return JENGA_OK;
    const int grid = grid_for((rows * (C / 8) + 255) / 256, 32768);
#define LAUNCH_GR(T)                                                                                              \
    hipLaunchKernelGGL(gate_residual_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)res, \
                       (const uint16_t*)y, (const uint16_t*)gate, (const uint16_t*)gate2, mask, (uint16_t*)out,   \
                       (long long)rows, (int)C, (long long)res_row_stride, (long long)y_row_stride,               \
                       (long long)o_row_stride)
    if (dtype == JENGA_BF16) LAUNCH_GR(BF16); else LAUNCH_GR(FP16);
#undef LAUNCH_GR
    JENGA_CHECK_LAUNCH("jenga_gate_residual");
    return JENGA_OK;
}

extern "C" int jenga_gelu_tanh(void* stream, const void* x, void* out, int64_t rows, int64_t C, int64_t x_row_stride,
                               int64_t o_row_stride, int dtype) {
    if (!x || !out || rows < 0 || C <= 0 || (C & 7) || (x_row_stride & 7) || (o_row_stride & 7)) {
        set_error("jenga_gelu_tanh: bad arguments");
        return JENGA_EINVAL;
    }
    if (dtype != JENGA_BF16 && dtype != JENGA_FP16) {
        set_error("jenga_gelu_tanh: dtype must be bf16 or fp16");
        return JENGA_EUNSUPPORTED;
    }
    if (rows == 0) return JENGA_OK;
    const int grid = grid_for((rows * (C / 8) + 255) / 256, 32768);
    if (dtype == JENGA_BF16)
        hipLaunchKernelGGL(gelu_tanh_kernel<BF16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x,
                           (uint16_t*)out, (long long)rows, (int)C, (long long)x_row_stride, (long long)o_row_stride);
    else
        hipLaunchKernelGGL(gelu_tanh_kernel<FP16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x,
                           (uint16_t*)out, (long long)rows, (int)C, (long long)x_row_stride, (long long)o_row_stride);
    JENGA_CHECK_LAUNCH("jenga_gelu_tanh");
    return JENGA_OK;
}

extern "C" int jenga_wan_ln_modulate(void* stream, const float* x, void* y, const float* weight, const float* bias,
                                     const float* shift, const float* scale, int64_t rows, int64_t C,
                                     int64_t x_row_stride, int64_t y_row_stride, float eps, int out_dtype,
                                     int round_ln) {
    if (!x || !y || rows < 0 || C <= 0 || (C & 3) || C > 6144 || (x_row_stride & 3) || (y_row_stride & 3) ||
        (!weight != !bias) || (!shift != !scale)) {
        set_error("jenga_wan_ln_modulate: bad arguments (C must be a multiple of 4 and <= 6144)");
        return JENGA_EINVAL;
    }
    if (out_dtype != JENGA_BF16 && out_dtype != JENGA_FP16) {
        set_error("jenga_wan_ln_modulate: out dtype must be bf16 or fp16");
        return JENGA_EUNSUPPORTED;
    }
    if (rows == 0) return JENGA_OK;
#define LAUNCH_WLM(T)                                                                                             \
    hipLaunchKernelGGL(wan_ln_modulate_kernel<T>, dim3(grid_for(rows, 65536)), dim3(256), 0, (hipStream_t)stream, \
                       x, (uint16_t*)y, weight, bias, shift, scale, (long long)rows, (int)C,                      \
                       (long long)x_row_stride, (long long)y_row_stride, eps, round_ln)
#define LAUNCH_WLMW(T, NV, A_, M_)                                                                               \
    hipLaunchKernelGGL((wan_ln_modulate_wave_kernel<T, NV, A_, M_>), dim3(grid_for((rows + 3) / 4, 65536)),       \
                       dim3(256), 0, (hipStream_t)stream, x, (uint16_t*)y, weight, bias, shift, scale,            \
                       (long long)rows, (long long)x_row_stride, (long long)y_row_stride, eps, round_ln)
#define LAUNCH_WLMW_T(T)                                                                                          \
    do {                                                                                                          \
        if (C == 5120) {                                                                                          \
            if (weight && shift) LAUNCH_WLMW(T, 20, true, true); else if (weight) LAUNCH_WLMW(T, 20, true, false); \
            else if (shift) LAUNCH_WLMW(T, 20, false, true); else LAUNCH_WLMW(T, 20, false, false);               \
        } else {                                                                                                  \
            if (weight && shift) LAUNCH_WLMW(T, 6, true, true); else if (weight) LAUNCH_WLMW(T, 6, true, false);   \
            else if (shift) LAUNCH_WLMW(T, 6, false, true); else LAUNCH_WLMW(T, 6, false, false);                 \
        }                                                                                                         \
    } while (0)
    if (C == 5120 || C == 1536) { if (out_dtype == JENGA_BF16) LAUNCH_WLMW_T(BF16); else LAUNCH_WLMW_T(FP16); }
    else if (out_dtype == JENGA_BF16) LAUNCH_WLM(BF16); else LAUNCH_WLM(FP16);
#undef LAUNCH_WLMW_T
#undef LAUNCH_WLMW
#undef LAUNCH_WLM
    JENGA_CHECK_LAUNCH("jenga_wan_ln_modulate");
    return JENGA_OK;
}

extern "C" int jenga_wan_gate_residual(void* stream, const float* x, const void* y, const float* gate, float* out,
                                       int64_t rows, int64_t C, int64_t x_row_stride, int64_t y_row_stride,
                                       int64_t o_row_stride, int y_dtype) {
    if (!x || !y || !out || rows < 0 || C <= 0 || (C & 7) || (x_row_stride & 3) || (y_row_stride & 7) ||
        (o_row_stride & 3)) {
        set_error("jenga_wan_gate_residual: bad arguments");
        return JENGA_EINVAL;
    }
    if (y_dtype != JENGA_BF16 && y_dtype != JENGA_FP16) {
        set_error("jenga_wan_gate_residual: y dtype must be bf16 or fp16");
        return JENGA_EUNSUPPORTED;
    }
    if (rows == 0) return JENGA_OK;
    const int grid = grid_for((rows * (C / 8) + 255) / 256, 32768);
#define LAUNCH_WGR(T)                                                                                             \
    hipLaunchKernelGGL(wan_gate_residual_kernel<T>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x,             \
                       (const uint16_t*)y, gate, out, (long long)rows, (int)C, (long long)x_row_stride,            \
                       (long long)y_row_stride, (long long)o_row_stride)
    if (y_dtype == JENGA_BF16) LAUNCH_WGR(BF16); else LAUNCH_WGR(FP16);
#undef LAUNCH_WGR
    JENGA_CHECK_LAUNCH("jenga_wan_gate_residual");
    return JENGA_OK;
}

extern "C" int jenga_stream_delay(void* stream, double microseconds) {
    if (!(microseconds >= 0.0) || microseconds > 5e6) {
        set_error("jenga_stream_delay: microseconds must be in [0, 5e6]");
        return JENGA_EINVAL;
    }
    int dev = 0, khz = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess ||
        khz <= 0) {
        set_error("jenga_stream_delay: cannot read the device's wall-clock rate");
        return JENGA_ELAUNCH;
    }
    const long long ticks = (long long)(microseconds * 1e-3 * (double)khz);
    if (ticks == 0) return JENGA_OK;
    hipLaunchKernelGGL(stream_delay_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ticks);
    JENGA_CHECK_LAUNCH("jenga_stream_delay");
    return JENGA_OK;
}
