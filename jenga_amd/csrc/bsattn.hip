// Block-sparse attention forward for gfx950 (CDNA4), head_dim 128, 128-token blocks.
//
// Work decomposition
//   workgroup = 256 threads = 4 waves = one 128-row query block of one (batch, head);
//   wave w owns query rows [32w, 32w+32); every lane owns ONE query row (q = lane&31), the two half-waves
//   (hi = lane>>5) split the key / head-dim index inside every MFMA.
//   KV is consumed in 64-key tiles (two per kept 128-block) staged through LDS and shared by the 4 waves:
//   K tile [64][128] (XOR-swizzled 16-B chunks), V tile pre-tiled by jenga_pack_v as [128 d][64 pos].
//
// MFMA formulation (v_mfma_f32_32x32x16_{bf16,f16}; C layout col = lane&31, row = (r&3)+8(r>>2)+4hi):
//   S^T[key][q] = sum_d K[key][d] Q[q][d]     A = K rows (ds_read_b128), B = Q (registers, loaded once)
//     -> lane holds the scores of ITS query row for 16 keys per 32-key group: softmax is lane-local
//        (31 fmax + one cross-half exchange per tile), no LDS round trip for P.
//   O^T[d][q]  += sum_k V[k][d] P[q][k]       A = V^T rows (ds_read_b128 of the pre-tiled image),
//                                             B = P packed to 16-bit straight from the S registers:
//     the key order inside a k-step is whatever the S registers hold (pv_key_of_pos) -- V was tiled to match,
//     so P never moves between lanes.
//   -> lane holds O[q][16 d per 32-d block]: the online-softmax rescale and the final 1/l are lane-local too.
//
// Numerics follow the reference Triton kernel (attention_block_triton_diffres.py:38-136): q is scaled by
// sm_scale*log2(e) and ROUNDED to the storage dtype before QK^T; scores/softmax in fp32 base 2; text_amp is
// added to the logits of kv blocks >= text_block_start; kv columns >= seqlen are -inf; P is rounded to the
// storage dtype before P.V; l accumulates the unrounded P; o = acc / l.  Text query blocks (TEXT=true) follow
// flash_attn_func instead (:371-380): unrounded q, scores * sm_scale, no mask, no amp.
#include "common.h"

namespace jenga {
namespace {

struct AttnParams {
    const uint16_t* q;
    const uint16_t* k;
    const uint16_t* vt;
    uint16_t* o;
    const int32_t* seqlens;
    const int32_t* idx;
    const int32_t* cnt;
    long long q_sb, q_ss, q_sh, k_sb, k_ss, k_sh, o_sb, o_ss, o_sh;
    int B, H, n_blocks, nq_img;
    int text_block_start;
    float qk_scale;   // sm_scale * log2(e), fp32
    float text_amp;
    int n_text_wg_pad;  // text workgroups (padded to a multiple of 8) come first in the grid
    int img_per_head;   // grid slots per (b,h) for image blocks (8*C with XCD remap, nq_img otherwise)
    int xcd_chunk;      // C, or 0 = plain order
};

constexpr int KT = 64;                      // keys per LDS tile
constexpr int K_TILE_BYTES = KT * 128 * 2;  // 16 KiB
constexpr int V_TILE_BYTES = 128 * KT * 2;  // 16 KiB
constexpr int BUF_BYTES = K_TILE_BYTES + V_TILE_BYTES;

// Direct global -> LDS staging (global_load_lds_dwordx4, 1 KiB per wave-instruction, no staging VGPRs, no
// ds_write).  The LDS image of a wave-instruction is lane-linear (base + lane*16), so the XOR swizzle of the tile is
// applied on the per-lane SOURCE address: piece pc = 4*wave + i covers K rows 4pc..4pc+3 (16 chunks each) or V^T rows
// 8pc..8pc+7 (8 chunks each); lane l lands on chunk c' = l&15 (l&7) of row 4pc + (l>>4) (8pc + (l>>3)) and therefore
// fetches source chunk c' ^ swizzle_key(row).
// Issued through inline asm: with the builtin, hipcc cannot prove that later ds_reads do not alias the DMA destination
// and drains vmcnt(0) before the first ds_read of the tile, which serialises prefetch and compute.  The asm form is
// invisible to its waitcnt bookkeeping, so the kernel waits itself (STAGE_WAIT) right before the barrier that
// publishes the tile.  M0 (DMA destination base) is saved/restored inside the statement (guide 5.7).
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}
#define GLDS16(GPTR, LDSOFF) glds16((const void*)(GPTR), (LDSOFF))
#define STAGE_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define STAGE_TILE(KPTR, VPTR, BUF)                                                               \
    do {                                                                                          \
        const uint16_t* kp_ = (KPTR);                                                             \
        const uint16_t* vp_ = (VPTR);                                                             \
        const unsigned lb_ = smem_base + (BUF) * BUF_BYTES + wave_u * 4096;                       \
        GLDS16(kp_ + k_src0, lb_);                                                                \
        GLDS16(kp_ + k_src1, lb_ + 1024);                                                         \
        GLDS16(kp_ + k_src2, lb_ + 2048);                                                         \
        GLDS16(kp_ + k_src3, lb_ + 3072);                                                         \
        GLDS16(vp_ + v_src0, lb_ + K_TILE_BYTES);                                                 \
        GLDS16(vp_ + v_src1, lb_ + K_TILE_BYTES + 1024);                                          \
        GLDS16(vp_ + v_src2, lb_ + K_TILE_BYTES + 2048);                                          \
        GLDS16(vp_ + v_src3, lb_ + K_TILE_BYTES + 3072);                                          \
    } while (0)

// Lazy running max: m~ is an integer-valued upper reference of each row's max, raised (by an integer step, so every
// rescale factor is an exact power of two) only when a row's new max exceeds it by more than LAZY_THR in log2 units.
// -m~ rides in the MFMA C operand of the first K.Q^T step, so S arrives already shifted and P = exp2(S) needs no
// subtraction; O and l are rescaled only in the (rare) raise path.  P <= 2^LAZY_THR keeps the storage dtype's relative
// precision (power-of-two scaling commutes with rounding), so results match the eager max up to fp32 rounding.
constexpr float LAZY_THR = 8.0f;

template <typename T, bool TEXT>
__device__ __forceinline__ void attn_block(const AttnParams& P, unsigned char* smem, int b, int h, int m) {
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 31, hi = lane >> 5;
    const int seqlen = P.seqlens ? P.seqlens[b] : P.n_blocks * 128;

    // ---- kept-block list ----
    const int32_t* list = nullptr;
    int nkept;
    if (TEXT) {
        nkept = P.n_blocks;
    } else {
        const long long row = ((long long)b * P.H + h) * P.nq_img + m;
        list = P.idx + row * P.n_blocks;
        nkept = P.cnt[row];
    }

    // ---- Q fragments: lane (q, hi) keeps Q[q][ds*16 + hi*8 .. +7] for ds = 0..7 ----
    const long long qrow = (long long)m * 128 + wave * 32 + lq;
    uint4 qf[8];
    {
        const uint16_t* qp = P.q + b * P.q_sb + qrow * P.q_ss + h * P.q_sh + hi * 8;
#pragma unroll
        for (int ds = 0; ds < 8; ++ds) {
            uint4 raw = *reinterpret_cast<const uint4*>(qp + ds * 16);
            if (!TEXT) {
                float f[8];
                unpack8<T>(raw, f);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = f[e] * P.qk_scale;
                raw = pack8<T>(f);
            }
            qf[ds] = raw;
        }
    }

    f32x16 oacc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
    float l_i = 0.f;
    float m_ref = 0.f;      // m~ (integer valued)
    f32x16 cinit;           // MFMA C operand of the first K.Q^T step: -m~ (TEXT: -m~ / qk_scale, scaled afterwards)
#pragma unroll
    for (int r = 0; r < 16; ++r) cinit[r] = 0.f;
    bool first = true;
    const float inv_scale = 1.0f / P.qk_scale;

    const uint16_t* kbh = P.k + b * P.k_sb + h * P.k_sh;
    const uint16_t* vbh = P.vt + ((long long)b * P.H + h) * (long long)P.n_blocks * 2 * (128 * KT);

    // ---- loop-invariant LDS addresses (buffer / group / d-block offsets are compile-time immediates) ----
    int k_addr[8], v_addr[4];
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) k_addr[ds] = lq * 256 + (((ds * 2 + hi) ^ (lq & 15)) << 4);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        v_addr[ks] = K_TILE_BYTES + lq * 128 + ((((ks >> 1) * 4 + hi * 2 + (ks & 1)) ^ ((lq >> 1) & 7)) << 4);
    // per-lane source offsets (elements) of the four K and four V^T pieces this wave stages per tile
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const unsigned smem_base =
        __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem);
    const int kr_ = 16 * wave_u + (lane >> 4), kc_ = lane & 15, ksw_ = lane >> 4;
    const long long k_src0 = (long long)(kr_ + 0) * P.k_ss + ((kc_ ^ (0 + ksw_)) << 3);
    const long long k_src1 = (long long)(kr_ + 4) * P.k_ss + ((kc_ ^ (4 + ksw_)) << 3);
    const long long k_src2 = (long long)(kr_ + 8) * P.k_ss + ((kc_ ^ (8 + ksw_)) << 3);
    const long long k_src3 = (long long)(kr_ + 12) * P.k_ss + ((kc_ ^ (12 + ksw_)) << 3);
    const int vr_ = 32 * wave_u + (lane >> 3), vc_ = lane & 7, vsw_ = lane >> 4;
    const int v_src0 = (vr_ + 0) * 64 + ((vc_ ^ ((0 + vsw_) & 7)) << 3);
    const int v_src1 = (vr_ + 8) * 64 + ((vc_ ^ ((4 + vsw_) & 7)) << 3);
    const int v_src2 = (vr_ + 16) * 64 + ((vc_ ^ ((8 + vsw_) & 7)) << 3);
    const int v_src3 = (vr_ + 24) * 64 + ((vc_ ^ ((12 + vsw_) & 7)) << 3);

    // one 64-key tile out of LDS buffer BUF (0/1); `blk` = kv block id, HALF = which half of it
#define COMPUTE_TILE(BUF, HALF)                                                                                  \
    do {                                                                                                         \
        const unsigned char* cur = smem + (BUF) * BUF_BYTES;                                                     \
        const int key0 = blk * 128 + (HALF) * KT;                                                                \
        if (TEXT || key0 < seqlen) { /* a tile entirely past seqlen contributes exp2(-inf) = 0 */                \
            f32x16 s0, s1;                                                                                       \
            {                                                                                                    \
                uint4 ka[8], kb[8];                                                                              \
                _Pragma("unroll") for (int ds = 0; ds < 8; ++ds) {                                               \
                    ka[ds] = *reinterpret_cast<const uint4*>(cur + k_addr[ds]);                                  \
                    kb[ds] = *reinterpret_cast<const uint4*>(cur + k_addr[ds] + 8192);                           \
                }                                                                                                \
                s0 = mfma32<T>(ka[0], qf[0], cinit);                                                             \
                s1 = mfma32<T>(kb[0], qf[0], cinit);                                                             \
                _Pragma("unroll") for (int ds = 1; ds < 8; ++ds) {                                               \
                    s0 = mfma32<T>(ka[ds], qf[ds], s0);                                                          \
                    s1 = mfma32<T>(kb[ds], qf[ds], s1);                                                          \
                }                                                                                                \
                /* issue order: LDS reads run 2 k-steps (4 reads) ahead of the MFMAs that consume them */        \
                __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);                                               \
                _Pragma("unroll") for (int g_ = 0; g_ < 5; ++g_) {                                               \
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                           \
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                           \
                }                                                                                                \
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);                                               \
            }                                                                                                    \
            if (TEXT) {                                                                                          \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                 \
                    s0[r] *= P.qk_scale;                                                                         \
                    s1[r] *= P.qk_scale;                                                                         \
                }                                                                                                \
            } else {                                                                                             \
                if (blk >= P.text_block_start) {                                                                 \
                    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                             \
                        s0[r] += P.text_amp;                                                                     \
                        s1[r] += P.text_amp;                                                                     \
                    }                                                                                            \
                }                                                                                                \
                if (key0 + KT > seqlen) {                                                                        \
                    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                             \
                        const int kk = key0 + (r & 3) + 8 * (r >> 2) + 4 * hi;                                   \
                        if (kk >= seqlen) s0[r] = -INFINITY;                                                     \
                        if (kk + 32 >= seqlen) s1[r] = -INFINITY;                                                \
                    }                                                                                            \
                }                                                                                                \
            }                                                                                                    \
            /* row max of the shifted scores (both half-waves of a row agree after the exchange) */              \
            float tmax = fmaxf(s0[0], s1[0]);                                                                    \
            _Pragma("unroll") for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, fmaxf(s0[r], s1[r]));              \
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32));                                                            \
            if (first || __any(tmax > LAZY_THR)) { /* raise m~ (rare): exact power-of-two rescale */             \
                const float delta = (first || tmax > LAZY_THR) ? ceilf(tmax) : 0.f;                              \
                const float f2 = __builtin_amdgcn_exp2f(-delta);                                                 \
                m_ref += delta;                                                                                  \
                l_i *= f2;                                                                                       \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                 \
                    s0[r] -= delta;                                                                              \
                    s1[r] -= delta;                                                                              \
                    cinit[r] = TEXT ? -m_ref * inv_scale : -m_ref;                                               \
                }                                                                                                \
                _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                    \
                    _Pragma("unroll") for (int r = 0; r < 16; ++r) oacc[i][r] *= f2;                             \
                first = false;                                                                                   \
            }                                                                                                    \
            float psum = 0.f;                                                                                    \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                     \
                s0[r] = __builtin_amdgcn_exp2f(s0[r]);                                                           \
                s1[r] = __builtin_amdgcn_exp2f(s1[r]);                                                           \
                psum += s0[r] + s1[r];                                                                           \
            }                                                                                                    \
            l_i += psum;                                                                                         \
            /* P -> 16-bit operands: k-step ks = 2g + s uses C registers 8s..8s+7 of group g */                  \
            uint4 pf[4];                                                                                         \
            pf[0] = make_uint4(pack2<T>(s0[0], s0[1]), pack2<T>(s0[2], s0[3]), pack2<T>(s0[4], s0[5]),           \
                               pack2<T>(s0[6], s0[7]));                                                          \
            pf[1] = make_uint4(pack2<T>(s0[8], s0[9]), pack2<T>(s0[10], s0[11]), pack2<T>(s0[12], s0[13]),       \
                               pack2<T>(s0[14], s0[15]));                                                        \
            pf[2] = make_uint4(pack2<T>(s1[0], s1[1]), pack2<T>(s1[2], s1[3]), pack2<T>(s1[4], s1[5]),           \
                               pack2<T>(s1[6], s1[7]));                                                          \
            pf[3] = make_uint4(pack2<T>(s1[8], s1[9]), pack2<T>(s1[10], s1[11]), pack2<T>(s1[12], s1[13]),       \
                               pack2<T>(s1[14], s1[15]));                                                        \
            /* O^T += V^T P^T: four independent accumulator chains per k-step */                                 \
            {                                                                                                    \
                uint4 va[4][4];                                                                                  \
                _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                 \
                    _Pragma("unroll") for (int db = 0; db < 4; ++db)                                             \
                        va[ks][db] = *reinterpret_cast<const uint4*>(cur + v_addr[ks] + db * 4096);              \
                _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                 \
                    _Pragma("unroll") for (int db = 0; db < 4; ++db)                                             \
                        oacc[db] = mfma32<T>(va[ks][db], pf[ks], oacc[db]);                                      \
                /* reads of k-step ks+1 are issued before the MFMAs of k-step ks */                              \
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);                                               \
                _Pragma("unroll") for (int g_ = 0; g_ < 2; ++g_) {                                               \
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                           \
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                                           \
                }                                                                                                \
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);                                               \
            }                                                                                                    \
        }                                                                                                        \
    } while (0)

    if (nkept > 0) {
        const int blk0 = TEXT ? 0 : list[0];
        STAGE_TILE(kbh + (long long)blk0 * 128 * P.k_ss, vbh + (long long)blk0 * 2 * (128 * KT), 0);
    }
    STAGE_WAIT();
    __syncthreads();

    for (int i = 0; i < nkept; ++i) {
        const int blk = TEXT ? i : list[i];
        // half 0 lives in buffer 0; fetch half 1 of the same block into buffer 1 meanwhile (buffer 1 was last read
        // before the barrier that ended the previous iteration)
        STAGE_TILE(kbh + ((long long)blk * 128 + KT) * P.k_ss, vbh + ((long long)blk * 2 + 1) * (128 * KT), 1);
        COMPUTE_TILE(0, 0);
        STAGE_WAIT();
        __syncthreads();
        {   // half 1 in buffer 1; fetch half 0 of the next kept block (clamped: the last re-fetch is never consumed)
            const int in = (i + 1 < nkept) ? i + 1 : i;
            const int nblk = TEXT ? in : list[in];
            STAGE_TILE(kbh + ((long long)nblk * 128) * P.k_ss, vbh + ((long long)nblk * 2) * (128 * KT), 0);
        }
        COMPUTE_TILE(1, 1);
        STAGE_WAIT();
        __syncthreads();
    }
#undef COMPUTE_TILE

    // ---- epilogue: o = acc / l, rows >= seqlen written as zeros (image rows only) ----
    const float l_tot = l_i + __shfl_xor(l_i, 32);
    const bool row_ok = TEXT || (qrow < seqlen);
    uint16_t* op = P.o + b * P.o_sb + qrow * P.o_ss + h * P.o_sh + hi * 4;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            uint2 w = make_uint2(0u, 0u);
            if (row_ok) {
                w.x = pack2<T>(__fdiv_rn(oacc[db][rq * 4 + 0], l_tot), __fdiv_rn(oacc[db][rq * 4 + 1], l_tot));
                w.y = pack2<T>(__fdiv_rn(oacc[db][rq * 4 + 2], l_tot), __fdiv_rn(oacc[db][rq * 4 + 3], l_tot));
            }
            *reinterpret_cast<uint2*>(op + db * 32 + rq * 8) = w;
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256, 2) bsattn_fwd_kernel(AttnParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int n_text = P.n_blocks - P.nq_img;
    const int id = blockIdx.x;
    if (id < P.n_text_wg_pad) {  // text query blocks first: longest work items start earliest
        if (id >= P.B * P.H * n_text) return;
        const int m = P.nq_img + id % n_text;
        const int bh = id / n_text;
        attn_block<T, true>(P, smem, bh / P.H, bh % P.H, m);
        return;
    }
    const int li = id - P.n_text_wg_pad;
    const int bh = li / P.img_per_head;
    const int r = li % P.img_per_head;
    int m;
    if (P.xcd_chunk) {  // workgroup id -> XCD is id % 8: give each XCD a contiguous range of query blocks
        m = (r & 7) * P.xcd_chunk + (r >> 3);
        if ((r >> 3) >= P.xcd_chunk || m >= P.nq_img) return;
    } else {
        m = r;
    }
    attn_block<T, false>(P, smem, bh / P.H, bh % P.H, m);
}

}  // namespace
}  // namespace jenga

using namespace jenga;

extern "C" int jenga_bsattn_fwd(void* stream, const void* q, const void* k, const void* vt, void* o,
                                const int32_t* seqlens, const int32_t* idx, const int32_t* cnt, int64_t B, int64_t H,
                                int64_t n_blocks, int64_t nq_img, int64_t q_sb, int64_t q_ss, int64_t q_sh,
                                int64_t k_sb, int64_t k_ss, int64_t k_sh, int64_t o_sb, int64_t o_ss, int64_t o_sh,
                                float sm_scale, float text_amp, int64_t text_block_start, int dtype, int flags) {
    if (!q || !k || !vt || !o || B <= 0 || H <= 0 || n_blocks <= 0 || nq_img < 0 || nq_img > n_blocks) {
        set_error("jenga_bsattn_fwd: bad arguments");
        return JENGA_EINVAL;
    }
    if (nq_img > 0 && (!idx || !cnt)) {
        set_error("jenga_bsattn_fwd: idx/cnt are required when nq_img > 0");
        return JENGA_EINVAL;
    }
    const int64_t strides[9] = {q_sb, q_ss, q_sh, k_sb, k_ss, k_sh, o_sb, o_ss, o_sh};
    for (int i = 0; i < 9; ++i)
        if (strides[i] & 7) {
            set_error("jenga_bsattn_fwd: strides must be multiples of 8 elements (16-byte rows)");
            return JENGA_EINVAL;
        }
    if (((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)vt & 15) || ((uintptr_t)o & 15)) {
        set_error("jenga_bsattn_fwd: pointers must be 16-byte aligned");
        return JENGA_EINVAL;
    }
    if (dtype != JENGA_BF16 && dtype != JENGA_FP16) {
        set_error("jenga_bsattn_fwd: dtype must be bf16 or fp16");
        return JENGA_EUNSUPPORTED;
    }
    AttnParams P;
    P.q = (const uint16_t*)q;
    P.k = (const uint16_t*)k;
    P.vt = (const uint16_t*)vt;
    P.o = (uint16_t*)o;
    P.seqlens = seqlens;
    P.idx = idx;
    P.cnt = cnt;
    P.q_sb = q_sb; P.q_ss = q_ss; P.q_sh = q_sh;
    P.k_sb = k_sb; P.k_ss = k_ss; P.k_sh = k_sh;
    P.o_sb = o_sb; P.o_ss = o_ss; P.o_sh = o_sh;
    P.B = (int)B; P.H = (int)H; P.n_blocks = (int)n_blocks; P.nq_img = (int)nq_img;
    P.text_block_start = (int)text_block_start;
    P.qk_scale = (float)((double)sm_scale * 1.44269504);
    P.text_amp = text_amp;
    const long long n_text = n_blocks - nq_img;
    const long long n_text_wg = B * H * n_text;
    P.n_text_wg_pad = (int)((n_text_wg + 7) / 8 * 8);
    if ((flags & JENGA_ATTN_XCD_REMAP) && nq_img >= 64) {
        P.xcd_chunk = (int)((nq_img + 7) / 8);
        P.img_per_head = P.xcd_chunk * 8;
    } else {
        P.xcd_chunk = 0;
        P.img_per_head = (int)nq_img;
    }
    const long long grid = (long long)P.n_text_wg_pad + B * H * (long long)P.img_per_head;
    if (grid <= 0 || grid > 0x7fffffffLL) {
        set_error("jenga_bsattn_fwd: grid size %lld out of range", grid);
        return JENGA_EINVAL;
    }
    const size_t smem = 2 * BUF_BYTES;
    hipError_t e;
    if (dtype == JENGA_BF16) {
        static bool attr_done = false;
        if (!attr_done) {
            (void)hipFuncSetAttribute((const void*)bsattn_fwd_kernel<BF16>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)smem);
            attr_done = true;
        }
        hipLaunchKernelGGL(bsattn_fwd_kernel<BF16>, dim3((unsigned)grid), dim3(256), smem, (hipStream_t)stream, P);
    } else {
        static bool attr_done16 = false;
        if (!attr_done16) {
            (void)hipFuncSetAttribute((const void*)bsattn_fwd_kernel<FP16>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)smem);
            attr_done16 = true;
        }
        hipLaunchKernelGGL(bsattn_fwd_kernel<FP16>, dim3((unsigned)grid), dim3(256), smem, (hipStream_t)stream, P);
    }
    e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("jenga_bsattn_fwd: %s", hipGetErrorString(e));
        return JENGA_ELAUNCH;
    }
    return JENGA_OK;
}
