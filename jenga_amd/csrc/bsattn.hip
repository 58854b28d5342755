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
// 4-wave kernel LDS: K ring of 3 slots (tile t in slot t % 3, fetched two tiles ahead) + V^T ring of 2 slots
// (fetched one tile ahead, first read in the second half of its tile): 80 KiB, two workgroups fill the CU's 160 KiB.
constexpr int W4_V_RING = 3 * K_TILE_BYTES;
constexpr int W4_LDS_BYTES = 3 * K_TILE_BYTES + 2 * V_TILE_BYTES;

// Direct global -> LDS staging (global_load_lds_dwordx4, 1 KiB per wave-instruction, no staging VGPRs, no
// ds_write).  The LDS image of a wave-instruction is lane-linear (base + lane*16), so the XOR swizzle of the tile is
// applied on the per-lane SOURCE address: piece pc = 4*wave + i covers K rows 4pc..4pc+3 (16 chunks each) or V^T rows
// 8pc..8pc+7 (8 chunks each); lane l lands on chunk c' = l&15 (l&7) of row 4pc + (l>>4) (8pc + (l>>3)) and therefore
// fetches source chunk c' ^ swizzle_key(row).
// Issued through inline asm: with the builtin, hipcc cannot prove that later ds_reads do not alias the DMA destination
// and drains vmcnt(0) before the first ds_read of the tile, which serialises prefetch and compute.  The asm form is
// invisible to its waitcnt bookkeeping, so the kernel waits itself (STAGE_WAIT) right before the barrier that
// publishes the tile.  One statement stages a whole tile: 4 K pieces + 4 V^T pieces of 1 KiB each for this wave,
// global address = uniform 64-bit base (SGPR pair) + per-lane 32-bit byte offset (loop-invariant VGPR), LDS
// destination = M0 (saved/restored inside the statement, guide 5.7), advanced by 1 KiB per piece.  The four V^T
// pieces go out BEFORE the four K pieces: loads complete in order, so `s_waitcnt vmcnt(4)` retires everything up to
// and including this call's V pieces while its K pieces (needed one tile later) stay in flight.
__device__ __forceinline__ void stage_tile(const void* kbase, const void* vbase, unsigned lds_k, unsigned lds_v,
                                           unsigned k0, unsigned k1, unsigned k2, unsigned k3, unsigned v0,
                                           unsigned v1, unsigned v2, unsigned v3) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %9, %4\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "global_load_lds_dwordx4 %10, %4\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "global_load_lds_dwordx4 %11, %4\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "global_load_lds_dwordx4 %12, %4\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %5, %3\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "global_load_lds_dwordx4 %6, %3\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "global_load_lds_dwordx4 %7, %3\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "global_load_lds_dwordx4 %8, %3\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds_k), "s"(lds_v), "s"(kbase), "s"(vbase), "v"(k0), "v"(k1), "v"(k2), "v"(k3), "v"(v0), "v"(v1),
          "v"(v2), "v"(v3)
        : "memory", "scc");
}
#define STAGE_TILE(KPTR, VPTR, KSLOT, VSLOT)                                                               \
    stage_tile((KPTR), (VPTR), smem_base + (KSLOT) * K_TILE_BYTES + wave_u * 4096,                        \
               smem_base + W4_V_RING + (VSLOT) * V_TILE_BYTES + wave_u * 4096, k_src0, k_src1, k_src2, k_src3, \
               v_src0, v_src1, v_src2, v_src3)
#define STAGE_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define STAGE_WAIT_KEEP_K() asm volatile("s_waitcnt vmcnt(4)" ::: "memory")

// Lazy running max: m~ is an integer-valued upper reference of each row's max, raised (by an integer step, so every
// rescale factor is an exact power of two) only when a row's new max exceeds it by more than LAZY_THR in log2 units.
// P = exp2(S - m~) costs one packed add per two scores; O and l are rescaled only in the (rare) raise path.
// P <= 2^LAZY_THR keeps the storage dtype's relative precision (power-of-two scaling commutes with rounding), so
// results match the eager max up to fp32 rounding.  Variants of this hot path that were measured and dropped (-m~ in
// the MFMA C operand, integer bf16 packing, integer exponent subtract, untied-C inline-asm MFMA) are in DESIGN.md §3.
constexpr float LAZY_THR = 8.0f;
constexpr float RAISE_SUM = 256.0f;   // 2^LAZY_THR
#define MFMA_PRIO(x) __builtin_amdgcn_s_setprio(x)
// s_waitcnt vmcnt(0) as a BUILTIN right behind the (rare) reload of the kept-list chunk: hipcc otherwise puts the wait
// for that load at the join in front of v_readlane, where it runs every block and drains the whole LDS-DMA prefetch
// (the hardware counter includes the asm loads the compiler knows nothing about).  Found in round 2 from the ISA.
#define LIST_LOAD_WAIT() __builtin_amdgcn_s_waitcnt(0x0F70)

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
        nkept = __builtin_amdgcn_readfirstlane(P.cnt[row]);
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
    float neg_m = 0.f;      // -m~ (integer valued)
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
    bool first = true;
    // output row pointer computed up front: lets the o_* strides die before the loop (SGPR pressure)
    uint16_t* const op = P.o + b * P.o_sb + qrow * P.o_ss + h * P.o_sh + hi * 4;

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
    const unsigned kss_b = (unsigned)P.k_ss * 2u;   // K row stride in bytes (64 rows x stride < 2^31, checked on the host)
    const unsigned k_src0 = (unsigned)(kr_ + 0) * kss_b + ((kc_ ^ (0 + ksw_)) << 4);
    const unsigned k_src1 = (unsigned)(kr_ + 4) * kss_b + ((kc_ ^ (4 + ksw_)) << 4);
    const unsigned k_src2 = (unsigned)(kr_ + 8) * kss_b + ((kc_ ^ (8 + ksw_)) << 4);
    const unsigned k_src3 = (unsigned)(kr_ + 12) * kss_b + ((kc_ ^ (12 + ksw_)) << 4);
    const int vr_ = 32 * wave_u + (lane >> 3), vc_ = lane & 7, vsw_ = lane >> 4;
    const unsigned v_src0 = (unsigned)(vr_ + 0) * 128 + ((vc_ ^ ((0 + vsw_) & 7)) << 4);
    const unsigned v_src1 = (unsigned)(vr_ + 8) * 128 + ((vc_ ^ ((4 + vsw_) & 7)) << 4);
    const unsigned v_src2 = (unsigned)(vr_ + 16) * 128 + ((vc_ ^ ((8 + vsw_) & 7)) << 4);
    const unsigned v_src3 = (unsigned)(vr_ + 24) * 128 + ((vc_ ^ ((12 + vsw_) & 7)) << 4);

    // one 64-key tile out of LDS buffer BUF (0/1); `blk` = kv block id, HALF = which half of it
#define COMPUTE_TILE(KSLOT, VSLOT, HALF, SLOW)                                                                   \
    do {                                                                                                         \
        const unsigned char* cur = smem + (KSLOT) * K_TILE_BYTES;                                                \
        const unsigned char* curv = smem + W4_V_RING + (VSLOT) * V_TILE_BYTES - K_TILE_BYTES;                    \
        const int key0 = blk * 128 + (HALF) * KT;                                                                \
        if (!(SLOW) || key0 < seqlen) { /* a tile entirely past seqlen contributes exp2(-inf) = 0 */             \
            f32x16 s0, s1;                                                                                       \
            float psum;                                                                                          \
            const bool pre = first; /* wave-uniform: update m~ from the row max BEFORE exponentiating */         \
            {                                                                                                    \
                {                                                                                                \
                    uint4 ka[8], kb[8];                                                                          \
                    MFMA_PRIO(1);                                                                                \
                    _Pragma("unroll") for (int ds = 0; ds < 8; ++ds) {                                           \
                        ka[ds] = *reinterpret_cast<const uint4*>(cur + k_addr[ds]);                              \
                        kb[ds] = *reinterpret_cast<const uint4*>(cur + k_addr[ds] + 8192);                       \
                    }                                                                                            \
                    s0 = mfma32<T>(ka[0], qf[0], zero16);                                                        \
                    s1 = mfma32<T>(kb[0], qf[0], zero16);                                                        \
                    _Pragma("unroll") for (int ds = 1; ds < 8; ++ds) {                                           \
                        s0 = mfma32<T>(ka[ds], qf[ds], s0);                                                      \
                        s1 = mfma32<T>(kb[ds], qf[ds], s1);                                                      \
                    }                                                                                            \
                    /* issue order: LDS reads run 3 k-steps (6 reads) ahead of the MFMAs that consume them */    \
                    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);                                           \
                    _Pragma("unroll") for (int g_ = 0; g_ < 5; ++g_) {                                           \
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                       \
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                       \
                    }                                                                                            \
                    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);                                           \
                    MFMA_PRIO(0);                                                                                \
                }                                                                                                \
                if (TEXT) {                                                                                      \
                    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                             \
                        s0[r] = s0[r] * P.qk_scale + neg_m;                                                      \
                        s1[r] = s1[r] * P.qk_scale + neg_m;                                                      \
                    }                                                                                            \
                } else if (SLOW) {                                                                               \
                    if (blk >= P.text_block_start) {                                                             \
                        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                         \
                            s0[r] += P.text_amp;                                                                 \
                            s1[r] += P.text_amp;                                                                 \
                        }                                                                                        \
                    }                                                                                            \
                    if (key0 + KT > seqlen) {                                                                    \
                        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                         \
                            const int kk = key0 + (r & 3) + 8 * (r >> 2) + 4 * hi;                               \
                            if (kk >= seqlen) s0[r] = -INFINITY;                                                 \
                            if (kk + 32 >= seqlen) s1[r] = -INFINITY;                                            \
                        }                                                                                        \
                    }                                                                                            \
                }                                                                                                \
                if (!TEXT) {                                                                                     \
                    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                             \
                        s0[r] += neg_m;                                                                          \
                        s1[r] += neg_m;                                                                          \
                    }                                                                                            \
                }                                                                                                \
                if (pre) { /* first tile of the row block */                                                     \
                    float tmax = fmaxf(s0[0], s1[0]);   /* scores are already relative to m~ */                  \
                    _Pragma("unroll") for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, fmaxf(s0[r], s1[r]));      \
                    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));                                                    \
                    const bool go_ = first ? (tmax > -1e20f) : (tmax > LAZY_THR);                                \
                    const float delta = go_ ? ceilf(tmax) : 0.f;                                                 \
                    const float f2 = __builtin_amdgcn_exp2f(-delta);                                             \
                    neg_m -= delta;                                                                              \
                    l_i *= f2;                                                                                   \
                    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                             \
                        s0[r] -= delta;                                                                          \
                        s1[r] -= delta;                                                                          \
                    }                                                                                            \
                    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                \
                        _Pragma("unroll") for (int r = 0; r < 16; ++r) oacc[i][r] *= f2;                         \
                    first = false;                                                                               \
                }                                                                                                \
                /* hot path: P = exp2(S - m~) and its row sum with packed adds, no row max.  P goes to its own   \
                   registers so that the raw scores survive until the check below. */                            \
                f32x16 p0, p1;                                                                                   \
                psum = 0.f;                                                                                      \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                 \
                    p0[r] = __builtin_amdgcn_exp2f(s0[r]);   /* v_exp_f32 hides under the partner's MFMAs */     \
                    p1[r] = __builtin_amdgcn_exp2f(s1[r]);                                                       \
                    psum += p0[r] + p1[r];                   /* the only FP-pipe work left on the hot path */    \
                }                                                                                                \
                /* rare: any P > 2^THR implies a row sum > 2^THR (and an overflowed exp2 gives inf): raise m~ by  \
                   an integer step from the row max of the intact scores, then exponentiate again.  Exact; the   \
                   row max is only ever computed in here and on the first tile. */                               \
                if (!pre && __any(!(psum <= RAISE_SUM))) {                                                       \
                    float tmax = fmaxf(s0[0], s1[0]);                                                            \
                    _Pragma("unroll") for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, fmaxf(s0[r], s1[r]));      \
                    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));                                                    \
                    /* once in here, re-reference whenever the row max is above m~ at all: with `> LAZY_THR` a    \
                       row carrying several scores 5..8 above m~ kept taking this branch on every tile */        \
                    const float delta = (tmax > 0.f) ? ceilf(tmax) : 0.f;                                        \
                    const float f2 = __builtin_amdgcn_exp2f(-delta);                                             \
                    neg_m -= delta;                                                                              \
                    l_i *= f2;                                                                                   \
                    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                \
                        _Pragma("unroll") for (int r = 0; r < 16; ++r) oacc[i][r] *= f2;                         \
                    psum = 0.f;                                                                                  \
                    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                             \
                        p0[r] = __builtin_amdgcn_exp2f(s0[r] - delta);                                           \
                        p1[r] = __builtin_amdgcn_exp2f(s1[r] - delta);                                           \
                        psum += p0[r] + p1[r];                                                                   \
                    }                                                                                            \
                }                                                                                                \
                s0 = p0;                                                                                         \
                s1 = p1;                                                                                         \
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
                MFMA_PRIO(1);                                                                                    \
                _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                 \
                    _Pragma("unroll") for (int db = 0; db < 4; ++db)                                             \
                        va[ks][db] = *reinterpret_cast<const uint4*>(curv + v_addr[ks] + db * 4096);             \
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
                MFMA_PRIO(0);                                                                                    \
            }                                                                                                    \
        }                                                                                                        \
    } while (0)

    // Kept lists ascend, so the blocks that need the text_amp add or the kv-length mask (text blocks, a padded last
    // block) are the LAST ones: the fast loop carries neither code path (the mask path alone costs 60+ SGPRs and made
    // the whole loop spill), the few remaining blocks go through the slow loop.
    int n_fast = nkept;
    if (!TEXT) {
        while (n_fast > 0) {
            const int bl = __builtin_amdgcn_readfirstlane(list[n_fast - 1]);
            if (bl >= P.text_block_start || (bl + 1) * 128 > seqlen) --n_fast; else break;
        }
    }
    // The kept list is read 64 entries at a time into one VGPR (lane j holds entry base+j) and entries are pulled out
    // with v_readlane: a per-iteration `list[i]` is a VECTOR load for hipcc (it cannot prove the list is not aliased
    // by the O stores), whose vmcnt(0) wait would also drain the LDS-DMA prefetch every iteration.
    int lchunk = 0;
#define LIST_GET(J, DST)                                                                                         \
    do {                                                                                                         \
        if (TEXT) {                                                                                              \
            DST = (J);                                                                                           \
        } else {                                                                                                 \
            if (((J) & 63) == 0) {                                                                               \
                lchunk = ((J) + lane < nkept) ? list[(J) + lane] : 0;                                            \
                LIST_LOAD_WAIT();   /* HERE, not at the join in front of v_readlane (every block: drains the DMA) */ \
            }                                                                                                    \
            DST = __builtin_amdgcn_readlane(lchunk, (J) & 63);                                                   \
        }                                                                                                        \
    } while (0)
    // Tile t (= 2*i + half of kept block i) reads K slot t % 3 and V^T slot t % 2.  At the start of tile t the DMA of
    // V(t+1) and K(t+2) goes out; at its end vmcnt(4) retires V(t+1) (and K(t+1), issued a tile earlier) and leaves
    // K(t+2) in flight across the barrier: K has two tiles of lead, V one and a half.
    int blk = 0, ks_cur = 0, ks_p2 = 2;   // K slot of the current tile / of tile t+2
    if (nkept > 0) {
        LIST_GET(0, blk);
        // prologue: V(0) + K(0), then (dummy V into the free slot) + K(1); keep K(1) in flight
        STAGE_TILE(kbh + (long long)blk * 128 * P.k_ss, vbh + (long long)blk * 2 * (128 * KT), 0, 0);
        STAGE_TILE(kbh + ((long long)blk * 128 + KT) * P.k_ss, vbh + ((long long)blk * 2 + 1) * (128 * KT), 1, 1);
    }
    STAGE_WAIT_KEEP_K();
    __syncthreads();

#define BLOCK_LOOP(FROM, TO, SLOW)                                                                               \
    for (int i = (FROM); i < (TO); ++i) {                                                                        \
        int nblk = blk;                                                                                          \
        if (i + 1 < nkept) LIST_GET(i + 1, nblk);                                                                \
        /* tile 2i: V(2i+1) = block i half 1 -> V slot 1; K(2i+2) = next block half 0 -> K slot (t+2)%3 */       \
        STAGE_TILE(kbh + ((long long)nblk * 128) * P.k_ss, vbh + ((long long)blk * 2 + 1) * (128 * KT), ks_p2, 1); \
        COMPUTE_TILE(ks_cur, 0, 0, SLOW);                                                                        \
        STAGE_WAIT_KEEP_K();                                                                                     \
        __syncthreads();                                                                                         \
        ks_cur = (ks_cur == 2) ? 0 : ks_cur + 1;                                                                 \
        ks_p2 = (ks_p2 == 2) ? 0 : ks_p2 + 1;                                                                    \
        /* tile 2i+1: V(2i+2) = next block half 0 -> V slot 0; K(2i+3) = next block half 1 */                    \
        STAGE_TILE(kbh + ((long long)nblk * 128 + KT) * P.k_ss, vbh + ((long long)nblk * 2) * (128 * KT), ks_p2, 0); \
        COMPUTE_TILE(ks_cur, 1, 1, SLOW);                                                                        \
        STAGE_WAIT_KEEP_K();                                                                                     \
        __syncthreads();                                                                                         \
        ks_cur = (ks_cur == 2) ? 0 : ks_cur + 1;                                                                 \
        ks_p2 = (ks_p2 == 2) ? 0 : ks_p2 + 1;                                                                    \
        blk = nblk;                                                                                              \
    }
    BLOCK_LOOP(0, n_fast, 0)
    if (!TEXT) {
        BLOCK_LOOP(n_fast, nkept, 1)
    }
    STAGE_WAIT();   // drain the last (unused) prefetch before the workgroup's LDS is released
#undef LIST_GET
#undef BLOCK_LOOP
#undef COMPUTE_TILE

    // ---- epilogue: o = acc / l, rows >= seqlen written as zeros (image rows only) ----
    const float l_tot = l_i + __shfl_xor(l_i, 32);
    const bool row_ok = TEXT || (qrow < seqlen);
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
                                const int32_t* seqlens, const int32_t* idx, const int32_t* cnt, const int32_t* order,
                                int64_t B, int64_t H, int64_t n_blocks, int64_t nq_img, int64_t q_sb, int64_t q_ss,
                                int64_t q_sh, int64_t k_sb, int64_t k_ss, int64_t k_sh, int64_t o_sb, int64_t o_ss,
                                int64_t o_sh, float sm_scale, float text_amp, int64_t text_block_start, int dtype,
                                int flags) {
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
    if (k_ss <= 0 || k_ss > (1 << 23)) {  // per-lane 32-bit byte offsets inside a 64-row K tile
        set_error("jenga_bsattn_fwd: key sequence stride %lld out of range", (long long)k_ss);
        return JENGA_EINVAL;
    }
    if (flags & JENGA_ATTN_LP)
        return jenga_bsattn_lp_launch(stream, q, k, vt, o, seqlens, idx, cnt, order, B, H, n_blocks, nq_img, q_sb, q_ss,
                                      q_sh, k_sb, k_ss, k_sh, o_sb, o_ss, o_sh, sm_scale, text_amp, text_block_start,
                                      dtype, flags);
    (void)order;   // a scheduling hint: the round-1 kernel below keeps its plain order
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
    hipError_t e;
    const size_t smem = W4_LDS_BYTES;
    static bool smem_set[2][64] = {};
    if (dtype == JENGA_BF16) {
        lp_set_smem_once((const void*)bsattn_fwd_kernel<BF16>, (int)smem, smem_set[0]);
        hipLaunchKernelGGL(bsattn_fwd_kernel<BF16>, dim3((unsigned)grid), dim3(256), smem, (hipStream_t)stream, P);
    } else {
        lp_set_smem_once((const void*)bsattn_fwd_kernel<FP16>, (int)smem, smem_set[1]);
        hipLaunchKernelGGL(bsattn_fwd_kernel<FP16>, dim3((unsigned)grid), dim3(256), smem, (hipStream_t)stream, P);
    }
    e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("jenga_bsattn_fwd: %s", hipGetErrorString(e));
        return JENGA_ELAUNCH;
    }
    return JENGA_OK;
}
