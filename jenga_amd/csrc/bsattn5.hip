// Block-sparse attention forward, fifth generation ("LQ": the LP pipeline on query-block PAIRS, 64 query rows per wave),
// gfx950, head_dim 128, 128-token blocks.  Same semantics as jenga_bsattn_fwd; lists come from jenga_pair_merge.
//
// Why: the LP kernel (bsattn3.hip) sits on two roofs at once -- 51 KB of memory-side
// traffic per kept (query block, kv block) pair = 7.4 TB/s, and a 1400 W board cap under which only energy per FLOP buys
// speed.  Both are per-BYTE costs of staging every 64 KB K/V block for ONE 128-row query block and of reading one 1-KiB LDS
// fragment per MFMA.  On the driver workload Hilbert-adjacent query blocks share 75 % of their kept kv blocks.  So:
//   * a workgroup = 4 waves = TWO Hilbert-adjacent query blocks A, B of one head (256 query rows), ONE workgroup per CU, one
//     wave per SIMD with the whole 512-entry register file.  Wave w owns rows [32w, 32w+32) of A AND of B, so all four
//     SIMDs carry the same load whatever the lists look like (a 64-contiguous-row split would idle the B waves on A-only
//     blocks).
//   * the two kept lists are merged beforehand (jenga_pair_merge) into [shared | A-only | B-only], each ascending.  A shared
//     kv block is staged ONCE for 256 query rows, and every K / V^T fragment read from LDS feeds TWO MFMAs (rows of A, rows of
//     B): half the staged bytes, half the LDS-DMA issues and half the fragment reads per FLOP of the LP kernel.  An A-only /
//     B-only block costs what it costs there (32-row items, lp_bb of lp_core.h).
//   * the kv order inside a row changes (this-row-only blocks first, then the shared ones, each ascending): online softmax
//     is order independent up to fp32 rounding of the running sums, and every rescale factor stays an exact power of two.
//     Results are deterministic (the order depends on the lists only), not bit-identical to the LP kernel's.
//   * the in-wave software pipeline, the LDS ring (K 3 + V^T 3 tiles of 64 keys, 96 KiB), the counted waits, the lazy integer
//     running max and the exact max-first path are the LP kernel's (lp_core.h); lq_bb below is its basic block for a 64-row
//     item = (32 rows of A + 32 rows of B) x 32 keys: 16 QK^T MFMAs + 16 P.V MFMAs + the softmax of 32 scores per lane in 32
//     fenced slots, one fragment read per TWO slots.
//   * text query blocks (TEXT mode: every kv block, no list) run as pairs of their own in the same launch, first in the grid.
// What round 2's pair kernel (experiments/bsattn2.hip, removed; git 882bacf) taught: with one wave per SIMD nothing covers
// a wave's LDS-DMA issue (60-185 cycles per 1-KiB piece), so the pieces-per-MFMA ratio is what has to fall -- that kernel
// halved the staged bytes of shared blocks but kept one fragment read and 8 pieces per 32 MFMAs; here a shared step carries
// 8 pieces per 64 MFMAs.  Dedicated loader waves are not an option at this register budget: the waves of a workgroup share
// one allocation, and a fifth 512-register wave has no SIMD to live on.
#include <cstdlib>

#define LP_QK_MFMA_ASM 1     // see lp_core.h: QK^T MFMAs with explicit register classes
#include "lp_balance.h"
#include "lp_core.h"

// elimination switches for A/B builds (tools/build_alt5.sh; results are WRONG with any of them, only the clock is read)
#ifndef LQ_X_NODMA
#define LQ_X_NODMA 0      // no LDS-DMA in the unrolled main loop
#endif
#ifndef LQ_X_NOBAR
#define LQ_X_NOBAR 0      // no vmcnt wait / barrier at the end of a step of the unrolled main loop
#endif
#ifndef LQ_AHEAD
#define LQ_AHEAD 4        // fragment reads run this many fragments (= twice as many MFMAs) ahead of their MFMAs
#endif
#ifndef LQ_DMA_SLOT0
#define LQ_DMA_SLOT0 3    // the four LDS-DMA pieces of a block go out in slots LQ_DMA_SLOT0 + 8 i
#endif
#ifndef LQ_WAIT_PAIR
#define LQ_WAIT_PAIR 0    // round 6: one counted lgkmcnt wait per FOUR MFMAs (two fragments) instead of one per two
#endif
#ifndef LQ_X_NOSM
#define LQ_X_NOSM 0       // no softmax arithmetic in lq_bb (P = const, no exact-path ballot)
#endif
#ifndef LQ_X_NOREAD
#define LQ_X_NOREAD 0     // no fragment reads in lq_bb (MFMAs on whatever the registers hold), no lgkmcnt waits
#endif

namespace jenga {
namespace {

struct LqParams {
    const uint16_t* q;
    const uint16_t* k;
    const uint16_t* vt;
    uint16_t* o;
    const int32_t* seqlens;
    const int32_t* pidx;   // [B,H,npair_img,n_blocks]: shared blocks, then A-only, then B-only (each ascending)
    const int32_t* pcnt;   // [B,H,npair_img,4]: n_shared, n_a, n_b, 0
    const int32_t* order;  // optional launch-order hint: position -> pair, per (b, h) (jenga_order_by_count on the pair work)
    long long q_sb, q_ss, q_sh, k_sb, k_ss, k_sh, o_sb, o_ss, o_sh;
    int B, H, n_blocks, nq_img;
    int npair_img, npair_txt;
    int text_block_start;
    float qk_scale;   // sm_scale * log2(e)
    float text_amp;
    int n_text_wg_pad;
    int img_per_head;
    int xcd_chunk;
    int bal_set;
};

// One basic block of the pair pipeline: lp_bb (lp_core.h) for a 64-row item.  Slot m < 16: QK^T MFMA on K fragment m >> 1
// for sub-block m & 1 (A, B); slot m >= 16: P.V MFMA on V^T fragment (m - 16) >> 1 for sub-block m & 1.  Fragment reads go
// out one per two slots, four fragments (= eight MFMAs) ahead, with one counted s_waitcnt lgkmcnt per two MFMAs.  The softmax
// of the previous item's 32 scores per lane (16 of A, 16 of B) is spread over the slots as in lp_bb.  Template parameters as
// lp_bb's.  PRE: 1 = this (HALF 0) block also requests the first four K fragments of the next block (HALF 1 of the same
// tile) under its last P.V MFMAs; 2 = this block's first four K fragments were requested that way.
template <typename T, bool TEXT, int HALF, bool DO_PV, bool DO_QK, bool DO_SM, int KOFF = -1, int VOFF = -1, int PRE = 0>
__device__ __forceinline__ void lq_bb(LpState& sa, LpState& sb, const unsigned char* kt, const unsigned char* vt,
                                      f32x16& snA, f32x16& snB, const f32x16& spA, const f32x16& spB,
                                      const uint4 (&poA)[2], const uint4 (&poB)[2], uint4 (&pnA)[2], uint4 (&pnB)[2],
                                      const int (&k_addr)[8], const int (&v_addr)[4], float qk_scale,
                                      uint4 (&frk)[8], const LpDma* dma = nullptr) {
    constexpr int KO = KOFF < 0 ? 0 : KOFF, VO = VOFF < 0 ? 0 : VOFF;
    static_assert(PRE == 0 || (DO_QK && DO_PV), "the cross-block fragment pipeline is for full blocks");
    static_assert(PRE != 1 || HALF == 0, "only the first half prefetches (the next tile may still be in flight)");
    constexpr int LAST = PRE == 1 ? 15 + LQ_AHEAD : (DO_PV ? 15 : 7);     // last fragment this block requests
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
    uint4 frv[8];
    if (LQ_X_NOREAD) {
#pragma unroll
        for (int f = 0; f < 8; ++f) frv[f] = frk[f];
    }
    float tt[32], xx[32];
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float halfA = 0.f, halfB = 0.f;
    uint32_t ww[16];
#define LQ_READ(F_)                                                                                                   \
    do {                                                                                                              \
        if (LQ_X_NOREAD) {                                                                                            \
        } else if ((F_) < 8) {                                                                                               \
            if (DO_QK) frk[(F_) & 7] = *reinterpret_cast<const uint4*>(kt + k_addr[(F_) & 7] + (HALF * 8192 + KO));   \
        } else if ((F_) < 16) {                                                                                       \
            if (DO_PV) frv[(F_) & 7] = *reinterpret_cast<const uint4*>(vt + v_addr[2 * HALF + (((F_) - 8) >> 2)] +    \
                                                                       ((((F_) - 8) & 3) * 4096 + VO));               \
        } else if ((F_) < 16 + LQ_AHEAD && PRE == 1) { /* K fragment (F_ - 16) of the next block: the other half of this tile */ \
            frk[(F_) & 7] = *reinterpret_cast<const uint4*>(kt + k_addr[(F_) & 7] + (8192 + KO));                     \
        }                                                                                                             \
    } while (0)
    /* score e = 0..31: sub-block e >> 4, C register e & 15.  Image rows: the scores are S - m~ already (C operand of the
       first MFMA), element e is exponentiated in slot X(e) = e / 2 for e < 4, e - 2 after that, added / packed one slot later,
       and the sum trees close in slot 31.  TEXT rows: scale-and-shift in slot e, exp2 in e + 1, add / pack in e + 2 (two
       softmax-only slots behind the last MFMA). */
#define LQ_SP(E_) (((E_) >> 4) ? spB[(E_) & 15] : spA[(E_) & 15])
#define LQ_SM_X(E_) ((E_) < 4 ? ((E_) >> 1) : (E_) - 2)
#define LQ_SM_ADD(E_)                                                                                                 \
    do {                                                                                                              \
        if (!LP_ROWSUM_DOT2) {                                                                                        \
            if (((E_) & 15) < 4) acc[((E_) >> 4) * 4 + ((E_) & 3)] = xx[E_];                                          \
            else acc[((E_) >> 4) * 4 + ((E_) & 3)] += xx[E_];                                                         \
        }                                                                                                             \
        if ((E_) & 1) {                                                                                               \
            ww[(E_) >> 1] = pack2<T>(xx[(E_) - 1], xx[E_]);                                                           \
            /* LP_ROWSUM_DOT2: one chain per sub-block, c + lo + hi of the packed pair (lp_core.h); in front of the pin */ \
            if (LP_ROWSUM_DOT2)                                                                                       \
                acc[((E_) >> 4) * 4] = lp_pair_sum<T>(ww[(E_) >> 1], ((E_) & 15) == 1 ? 0.f : acc[((E_) >> 4) * 4]);  \
            asm volatile("" : "+v"(ww[(E_) >> 1]));   /* stay in this slot */                                         \
        }                                                                                                             \
    } while (0)
#define LQ_SM(M_)                                                                                                     \
    do {                                                                                                              \
        if (DO_SM && !LQ_X_NOSM) {                                                                                    \
            if (TEXT) {                                                                                               \
                if ((M_) >= 2 && (M_) < 34) LQ_SM_ADD(((M_) - 2) & 31);                                               \
                if ((M_) >= 1 && (M_) < 33) xx[((M_) - 1) & 31] = __builtin_amdgcn_exp2f(tt[((M_) - 1) & 31]);        \
                if ((M_) < 32) tt[(M_) & 31] = LQ_SP((M_) & 31) * qk_scale + ((((M_) & 31) >> 4) ? sb.neg_m : sa.neg_m); \
                if ((M_) == 33) {                                                                                     \
                    halfA = LP_ROWSUM_DOT2 ? acc[0] : (acc[0] + acc[1]) + (acc[2] + acc[3]);                          \
                    halfB = LP_ROWSUM_DOT2 ? acc[4] : (acc[4] + acc[5]) + (acc[6] + acc[7]);                          \
                }                                                                                                     \
            } else {                                                                                                  \
                _Pragma("unroll") for (int e_ = 0; e_ < 32; ++e_)                                                     \
                    if (LQ_SM_X(e_) + 1 == (M_)) LQ_SM_ADD(e_);                                                       \
                _Pragma("unroll") for (int e_ = 0; e_ < 32; ++e_)                                                     \
                    if (LQ_SM_X(e_) == (M_)) xx[e_] = __builtin_amdgcn_exp2f(LQ_SP(e_));                              \
                if ((M_) == 15) halfA = LP_ROWSUM_DOT2 ? acc[0] : (acc[0] + acc[1]) + (acc[2] + acc[3]);              \
                if ((M_) == 31) halfB = LP_ROWSUM_DOT2 ? acc[4] : (acc[4] + acc[5]) + (acc[6] + acc[7]);              \
            }                                                                                                         \
        }                                                                                                             \
    } while (0)
#define LQ_LGKM(N_)                                                                                                   \
    do {                                                                                                              \
        if ((N_) == 1) __builtin_amdgcn_s_waitcnt(0xC17F);                                                            \
        else if ((N_) == 2) __builtin_amdgcn_s_waitcnt(0xC27F);                                                       \
        else if ((N_) == 3) __builtin_amdgcn_s_waitcnt(0xC37F);                                                       \
        else if ((N_) == 4) __builtin_amdgcn_s_waitcnt(0xC47F);                                                       \
        else if ((N_) == 5) __builtin_amdgcn_s_waitcnt(0xC57F);                                                       \
        else if ((N_) == 6) __builtin_amdgcn_s_waitcnt(0xC67F);                                                       \
        else if ((N_) == 7) __builtin_amdgcn_s_waitcnt(0xC77F);                                                       \
        else __builtin_amdgcn_s_waitcnt(0xC07F);                                                                      \
    } while (0)
    /* slot m, m even, issues the MFMAs m and m + 1 on fragment m >> 1: that fragment must be there; the reads requested
       behind it are fragments (m >> 1) + 1 .. min((m >> 1) + 3, LAST) */
#define LQ_SLOT(M_)                                                                                                   \
    do {                                                                                                              \
        if (!LQ_X_NOREAD && !((M_) & 1) && (((M_) < 16 && DO_QK) || ((M_) >= 16 && DO_PV))) {                         \
            if (!LQ_WAIT_PAIR) {                                                                                      \
                LQ_LGKM(LAST - ((M_) >> 1) < LQ_AHEAD - 1 ? LAST - ((M_) >> 1) : LQ_AHEAD - 1);                       \
                __builtin_amdgcn_sched_barrier(0);   /* or hipcc moves the MFMA above the wait and adds its own */    \
            } else if (((M_) & 3) == 0) {                                                                             \
                /* one wait per FOUR MFMAs: fragments f = m >> 1 and f + 1 must both be there; requested behind */    \
                /* them: f + 2 .. min(f + LQ_AHEAD - 1, LAST) */                                                      \
                LQ_LGKM(LAST - ((M_) >> 1) - 1 < LQ_AHEAD - 2 ? (LAST - ((M_) >> 1) - 1 < 0 ? 0 : LAST - ((M_) >> 1) - 1) \
                                                              : LQ_AHEAD - 2);                                        \
                __builtin_amdgcn_sched_barrier(0);                                                                    \
            }                                                                                                         \
        }                                                                                                             \
        if ((M_) < 16) {                                                                                              \
            if (DO_QK) {                                                                                              \
                if ((M_) & 1) {                                                                                       \
                    if ((M_) == 1) { if (TEXT) lp_qk_zero<T>(snB, frk[0], sb.qf[0], zero16); else lp_qk_first<T>(snB, frk[0], sb.qf[0], sb.cinit); } \
                    else lp_qk_acc<T>(snB, frk[((M_) >> 1) & 7], sb.qf[((M_) >> 1) & 7]);                             \
                } else {                                                                                              \
                    if ((M_) == 0) { if (TEXT) lp_qk_zero<T>(snA, frk[0], sa.qf[0], zero16); else lp_qk_first<T>(snA, frk[0], sa.qf[0], sa.cinit); } \
                    else lp_qk_acc<T>(snA, frk[((M_) >> 1) & 7], sa.qf[((M_) >> 1) & 7]);                             \
                }                                                                                                     \
            }                                                                                                         \
        } else if (DO_PV) {                                                                                           \
            if ((M_) & 1)                                                                                             \
                sb.o[(((M_) - 16) >> 1) & 3] = mfma32<T>(frv[(((M_) - 16) >> 1) & 7], poB[((M_) - 16) >> 3],          \
                                                         sb.o[(((M_) - 16) >> 1) & 3]);                               \
            else                                                                                                      \
                sa.o[(((M_) - 16) >> 1) & 3] = mfma32<T>(frv[(((M_) - 16) >> 1) & 7], poA[((M_) - 16) >> 3],          \
                                                         sa.o[(((M_) - 16) >> 1) & 3]);                               \
        }                                                                                                             \
        if (((M_) & 1) == 0 && !(!DO_QK && ((M_) >> 1) + LQ_AHEAD < 8 + LQ_AHEAD)) {                                  \
            LQ_READ(((M_) >> 1) + LQ_AHEAD);                                                                          \
        }                                                                                                             \
        if (dma && !LQ_X_NODMA) {                                                                                     \
            if ((M_) == LQ_DMA_SLOT0) lp_stage1<0>(*dma);                                                             \
            if ((M_) == LQ_DMA_SLOT0 + 8) lp_stage1<1>(*dma);                                                         \
            if ((M_) == LQ_DMA_SLOT0 + 16) lp_stage1<2>(*dma);                                                        \
            if ((M_) == LQ_DMA_SLOT0 + 24) lp_stage1<3>(*dma);                                                        \
        }                                                                                                             \
        LQ_SM(M_);                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
    } while (0)
    if (DO_QK && PRE != 2) {
        LQ_READ(0); LQ_READ(1); LQ_READ(2); LQ_READ(3);
        if (LQ_AHEAD > 4) { LQ_READ(4); }
        if (LQ_AHEAD > 5) { LQ_READ(5); }
        if (LQ_AHEAD > 6) { LQ_READ(6); }
    }
    if (!DO_QK) {
        LQ_READ(8); LQ_READ(9); LQ_READ(10); LQ_READ(11);
        if (LQ_AHEAD > 4) { LQ_READ(12); }
        if (LQ_AHEAD > 5) { LQ_READ(13); }
        if (LQ_AHEAD > 6) { LQ_READ(14); }
    }
    __builtin_amdgcn_sched_barrier(0);
    LQ_SLOT(0); LQ_SLOT(1); LQ_SLOT(2); LQ_SLOT(3); LQ_SLOT(4); LQ_SLOT(5); LQ_SLOT(6); LQ_SLOT(7);
    LQ_SLOT(8); LQ_SLOT(9); LQ_SLOT(10); LQ_SLOT(11); LQ_SLOT(12); LQ_SLOT(13); LQ_SLOT(14); LQ_SLOT(15);
    LQ_SLOT(16); LQ_SLOT(17); LQ_SLOT(18); LQ_SLOT(19); LQ_SLOT(20); LQ_SLOT(21); LQ_SLOT(22); LQ_SLOT(23);
    LQ_SLOT(24); LQ_SLOT(25); LQ_SLOT(26); LQ_SLOT(27); LQ_SLOT(28); LQ_SLOT(29); LQ_SLOT(30); LQ_SLOT(31);
    LQ_SM(32);
    LQ_SM(33);
#undef LQ_READ
#undef LQ_SP
#undef LQ_SM_X
#undef LQ_SM_ADD
#undef LQ_SM
#undef LQ_LGKM
#undef LQ_SLOT
    if (DO_QK && !DO_PV) lp_qk_settle();
    if (DO_SM && LQ_X_NOSM) {
        pnA[0] = pnA[1] = pnB[0] = pnB[1] = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
        sa.l += 32.f;
        sb.l += 32.f;
    } else if (DO_SM) {
        const auto swA = __builtin_amdgcn_permlane32_swap(__float_as_uint(halfA), __float_as_uint(halfA), false, false);
        const auto swB = __builtin_amdgcn_permlane32_swap(__float_as_uint(halfB), __float_as_uint(halfB), false, false);
        float psumA = __uint_as_float(swA[0]) + __uint_as_float(swA[1]);   // both half-lanes: the row's 32 keys
        float psumB = __uint_as_float(swB[0]) + __uint_as_float(swB[1]);
        pnA[0] = make_uint4(ww[0], ww[1], ww[2], ww[3]);
        pnA[1] = make_uint4(ww[4], ww[5], ww[6], ww[7]);
        pnB[0] = make_uint4(ww[8], ww[9], ww[10], ww[11]);
        pnB[1] = make_uint4(ww[12], ww[13], ww[14], ww[15]);
        if ((__builtin_amdgcn_ballot_w64(!(psumA <= LP_RAISE_SUM)) |
             __builtin_amdgcn_ballot_w64(sa.l + psumA < lp_tiny<T>())) != 0ull)
            lp_exact<T, TEXT>(sa, spA, pnA, psumA, qk_scale, DO_QK ? &snA : nullptr);
        sa.l += psumA;
        if ((__builtin_amdgcn_ballot_w64(!(psumB <= LP_RAISE_SUM)) |
             __builtin_amdgcn_ballot_w64(sb.l + psumB < lp_tiny<T>())) != 0ull)
            lp_exact<T, TEXT>(sb, spB, pnB, psumB, qk_scale, DO_QK ? &snB : nullptr);
        sb.l += psumB;
    }
}

#define LQ_WAIT_KEEP8() asm volatile("s_waitcnt vmcnt(8)" ::: "memory")

template <int M> struct LqMode { static constexpr int value = M; };   // 0: rows of A only, 1: rows of B only, 2: both

__device__ __forceinline__ void lq_state_init(LpState& st) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) st.o[i][r] = 0.f;
    st.l = 0.f;
    st.neg_m = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) st.cinit[r] = 0.f;
    lp_cinit_pin(st.cinit);
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) st.qf[ds] = lp_u32x4{0u, 0u, 0u, 0u};
}

// query blocks mA (A) and mA + 1 (B, when has_b) of head (b, h); image pairs: `list` = [shared | A-only | B-only]
template <typename T, bool TEXT>
__device__ __forceinline__ void attn_pair_lq(const LqParams& P, unsigned char* smem, int b, int h, int mA, bool has_b,
                                             const int32_t* list, int n_sh, int n_a, int n_b) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 31, hi = lane >> 5;
    const int seqlen = P.seqlens ? __builtin_amdgcn_readfirstlane(P.seqlens[b]) : P.n_blocks * 128;

    LpState sa, sb;
    lq_state_init(sa);
    lq_state_init(sb);
    const long long qrowA = (long long)mA * 128 + wave_u * 32 + lq;
    const long long qrowB = qrowA + 128;
    auto load_q = [&](LpState& st, long long qrow) {
        const uint16_t* qp = P.q + b * P.q_sb + qrow * P.q_ss + h * P.q_sh + hi * 8;
#pragma unroll
        for (int ds = 0; ds < 8; ++ds) {
            uint4 raw = *reinterpret_cast<const uint4*>(qp + ds * 16);
            if (!TEXT) {   // q~ = dtype(q * sm_scale * log2 e)   (reference :87-88)
                float f[8];
                unpack8<T>(raw, f);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = f[e] * P.qk_scale;
                raw = pack8<T>(f);
            }
            st.qf[ds] = __builtin_bit_cast(lp_u32x4, raw);
            asm volatile("" : "+a"(st.qf[ds]));     // a tuple in the accumulator half from here on
        }
    };
    load_q(sa, qrowA);
    // (no `if (has_b)`: a Q tuple that is a merge of "loaded" and "zero" is no longer the asm statement's AGPR output, the
    // allocator keeps it in VGPRs and copies it to AGPRs right in front of every MFMA -- a hazard nobody covers for an asm
    // MFMA (tools/isa_hazards.py); a missing B computes on A's rows and is not stored)
    load_q(sb, has_b ? qrowB : qrowA);

    const uint16_t* kbh = P.k + b * P.k_sb + h * P.k_sh;
    const uint16_t* vbh = P.vt + ((long long)b * P.H + h) * (long long)P.n_blocks * (2 * 128 * 64);

    int k_addr[8], v_addr[4];
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) k_addr[ds] = LP_K_RING + lq * 256 + (((ds * 2 + hi) ^ (lq & 15)) << 4);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        v_addr[ks] = LP_V_RING + lq * 128 + ((((ks >> 1) * 4 + hi * 2 + (ks & 1)) ^ ((lq >> 1) & 7)) << 4);
    const unsigned smem_base =
        __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem);
    const int kr_ = 16 * wave_u + (lane >> 4), kc_ = lane & 15, ksw_ = lane >> 4;
    const unsigned kss_b = (unsigned)P.k_ss * 2u;
    const unsigned k_src0 = (unsigned)(kr_ + 0) * kss_b + ((kc_ ^ (0 + ksw_)) << 4);
    const unsigned k_src1 = (unsigned)(kr_ + 4) * kss_b + ((kc_ ^ (4 + ksw_)) << 4) - 1024u;
    const unsigned k_src2 = (unsigned)(kr_ + 8) * kss_b + ((kc_ ^ (8 + ksw_)) << 4) - 2048u;
    const unsigned k_src3 = (unsigned)(kr_ + 12) * kss_b + ((kc_ ^ (12 + ksw_)) << 4) - 3072u;
    const int vr_ = 32 * wave_u + (lane >> 3), vc_ = lane & 7, vsw_ = lane >> 4;
    const unsigned v_src0 = (unsigned)(vr_ + 0) * 128 + ((vc_ ^ ((0 + vsw_) & 7)) << 4);
    const unsigned v_src1 = (unsigned)(vr_ + 8) * 128 + ((vc_ ^ ((4 + vsw_) & 7)) << 4) - 1024u;
    const unsigned v_src2 = (unsigned)(vr_ + 16) * 128 + ((vc_ ^ ((8 + vsw_) & 7)) << 4) - 2048u;
    const unsigned v_src3 = (unsigned)(vr_ + 24) * 128 + ((vc_ ^ ((12 + vsw_) & 7)) << 4) - 3072u;

    // One walk = the LP pipeline over one list segment `lst[0 .. nkept)` for the rows of A (MODE 0), of B (1) or of both (2):
    // prologue, steps, drain, then the tail blocks that need text_amp / the kv-length mask in the unpipelined form.
    auto walk = [&](auto mode_tag, const int32_t* lst, const int nkept) {
        constexpr int MODE = decltype(mode_tag)::value;
        if (nkept <= 0) return;
        LpState& s1 = MODE == 1 ? sb : sa;     // the single sub-block of modes 0 / 1
        int lchunk = 0, lbase = -64;
        auto blk_at = [&](int i) -> int {
            if (i >= nkept) i = nkept - 1;   // the last steps stage one (unused) tile more: same piece count every step
            if (TEXT) return i;
            if (i < lbase || i >= lbase + 64) {
                lbase = i & ~63;
                lchunk = (lbase + lane < nkept) ? lst[lbase + lane] : 0;
                __builtin_amdgcn_s_waitcnt(0x0F70);   // wait HERE for the (rare) reload (bsattn3.hip)
            }
            return __builtin_amdgcn_readlane(lchunk, i - lbase);
        };
        auto issue_k_at = [&](int t, int slot) {
            const int tc = t < 2 * nkept ? t : 2 * nkept - 1;
            const int blk = blk_at(tc >> 1);
            lp_stage4(reinterpret_cast<const unsigned char*>(kbh) +
                          (unsigned long long)((unsigned)blk * 128u + (unsigned)(tc & 1) * 64u) * kss_b,
                      smem_base + LP_K_RING + slot * LP_TILE + wave_u * 4096, k_src0, k_src1, k_src2, k_src3);
        };
        auto issue_v_at = [&](int t, int slot) {
            const int tc = t < 2 * nkept ? t : 2 * nkept - 1;
            const int blk = blk_at(tc >> 1);
            lp_stage4(reinterpret_cast<const unsigned char*>(vbh) +
                          (unsigned long long)((unsigned)blk * 2u + (unsigned)(tc & 1)) * (128u * 64u * 2u),
                      smem_base + LP_V_RING + slot * LP_TILE + wave_u * 4096, v_src0, v_src1, v_src2, v_src3);
        };
        auto lp_window = [&](int t) {
            if (TEXT) return;
            const int first = t >> 1, last = (t + 7) >> 1;
            if (first < lbase || last >= lbase + 64) {
                lbase = first;
                lchunk = (lbase + lane < nkept) ? lst[lbase + lane] : 0;
                __builtin_amdgcn_s_waitcnt(0x0F70);
            }
        };
        auto blk_fast = [&](int i) -> int {
            if (i >= nkept) i = nkept - 1;
            if (TEXT) return i;
            return __builtin_amdgcn_readlane(lchunk, i - lbase);
        };
        auto desc_k_at = [&](int t, int slot) {
            const int tc = t < 2 * nkept ? t : 2 * nkept - 1;
            const int blk = blk_fast(tc >> 1);
            LpDma d;
            d.base = reinterpret_cast<const unsigned char*>(kbh) +
                     (unsigned long long)((unsigned)blk * 128u + (unsigned)(tc & 1) * 64u) * kss_b;
            d.lds = smem_base + LP_K_RING + slot * LP_TILE + wave_u * 4096;
            d.o[0] = k_src0; d.o[1] = k_src1; d.o[2] = k_src2; d.o[3] = k_src3;
            return d;
        };
        auto desc_v_at = [&](int t, int slot) {
            const int tc = t < 2 * nkept ? t : 2 * nkept - 1;
            const int blk = blk_fast(tc >> 1);
            LpDma d;
            d.base = reinterpret_cast<const unsigned char*>(vbh) +
                     (unsigned long long)((unsigned)blk * 2u + (unsigned)(tc & 1)) * (128u * 64u * 2u);
            d.lds = smem_base + LP_V_RING + slot * LP_TILE + wave_u * 4096;
            d.o[0] = v_src0; d.o[1] = v_src1; d.o[2] = v_src2; d.o[3] = v_src3;
            return d;
        };
        auto issue_k = [&](int t) { issue_k_at(t, t % 3); };
        auto issue_v = [&](int t) { issue_v_at(t, t % 3); };
        auto kslot = [&](int t) { return smem + (t % 3) * LP_TILE; };   // + k_addr (LP_K_RING inside)
        auto vslot = [&](int t) { return smem + (t % 3) * LP_TILE; };   // + v_addr (LP_V_RING inside)

        // blocks at the tail of the ascending segment that need the text_amp / kv-length path
        int n_fast = nkept;
        if (!TEXT) {
            while (n_fast > 0) {
                const int bl = blk_at(n_fast - 1);
                if (bl >= P.text_block_start || (bl + 1) * 128 > seqlen) --n_fast; else break;
            }
        }
        // tiles lying entirely behind the kv length contribute exp2(-inf) = 0: not staged at all
        int t_all = 2 * nkept;
        if (!TEXT) {
            while (t_all > 0 && blk_at((t_all - 1) >> 1) * 128 + ((t_all - 1) & 1) * 64 >= seqlen) --t_all;
        }
        const int t_fast = 2 * n_fast < t_all ? 2 * n_fast : t_all;

        f32x16 sXa, sXb, sYa, sYb;
        uint4 pXa[2], pXb[2], pYa[2], pYb[2];
        uint4 frk[8];
#pragma unroll
        for (int r = 0; r < 16; ++r) sXa[r] = sXb[r] = sYa[r] = sYb[r] = 0.f;
        pXa[0] = pXa[1] = pXb[0] = pXb[1] = pYa[0] = pYa[1] = pYb[0] = pYb[1] = make_uint4(0u, 0u, 0u, 0u);

        // prologue: K(0), K(1), V^T(0)
        issue_k(0);
        issue_k(1);
        issue_v(0);
        LP_WAIT_ALL();
        __syncthreads();

        // step t: stage V^T(t+1) and K(t+2); QK^T on K(t); P.V on V^T(t-1).  Both rings have three slots here (96 KiB; one
        // workgroup per CU): what a step stages is needed one (V^T) / two (K) steps later, so the wait at the end of a step
        // leaves ALL of the step's own eight pieces in flight -- the LP kernel's V^T(t), staged in the step that ends with
        // waiting for it, is covered there by the CU's other workgroup; with one wave per SIMD nothing would cover it
#define LQ_STEP(T_, PV_, SM0_)                                                                                        \
    do {                                                                                                              \
        issue_v((T_) + 1);                                                                                            \
        if constexpr (MODE == 2)                                                                                      \
            lq_bb<T, TEXT, 0, PV_, true, SM0_>(sa, sb, kslot(T_), vslot((T_) - 1), sXa, sXb, sYa, sYb, pXa, pXb, pYa, \
                                               pYb, k_addr, v_addr, P.qk_scale, frk);                                 \
        else                                                                                                          \
            lp_bb<T, TEXT, 0, PV_, true, SM0_>(s1, kslot(T_), vslot((T_) - 1), sXa, sYa, pXa, pYa, k_addr, v_addr,    \
                                               P.qk_scale, frk);                                                      \
        issue_k((T_) + 2);                                                                                            \
        if constexpr (MODE == 2)                                                                                      \
            lq_bb<T, TEXT, 1, PV_, true, true>(sa, sb, kslot(T_), vslot((T_) - 1), sYa, sYb, sXa, sXb, pYa, pYb, pXa, \
                                               pXb, k_addr, v_addr, P.qk_scale, frk);                                 \
        else                                                                                                          \
            lp_bb<T, TEXT, 1, PV_, true, true>(s1, kslot(T_), vslot((T_) - 1), sYa, sXa, pYa, pXa, k_addr, v_addr,    \
                                               P.qk_scale, frk);                                                      \
        LQ_WAIT_KEEP8();                                                                                              \
        __syncthreads();                                                                                              \
    } while (0)
        // step t0 + J of the unrolled loop of the shared segment, t0 = 1 (mod 6): every ring slot a compile-time constant
#define LQ_STEP_C(T0_, J_)                                                                                            \
    do {                                                                                                              \
        {                                                                                                             \
            const LpDma dv_ = desc_v_at((T0_) + (J_) + 1, (2 + (J_)) % 3);                                            \
            lq_bb<T, TEXT, 0, true, true, true, ((1 + (J_)) % 3) * LP_TILE, ((J_) % 3) * LP_TILE, 1>(                 \
                sa, sb, smem, smem, sXa, sXb, sYa, sYb, pXa, pXb, pYa, pYb, k_addr, v_addr, P.qk_scale, frk, &dv_);   \
        }                                                                                                             \
        {                                                                                                             \
            const LpDma dk_ = desc_k_at((T0_) + (J_) + 2, (J_) % 3);                                                  \
            lq_bb<T, TEXT, 1, true, true, true, ((1 + (J_)) % 3) * LP_TILE, ((J_) % 3) * LP_TILE, 2>(                 \
                sa, sb, smem, smem, sYa, sYb, sXa, sXb, pYa, pYb, pXa, pXb, k_addr, v_addr, P.qk_scale, frk, &dk_);   \
        }                                                                                                             \
        if (!LQ_X_NOBAR) {                                                                                            \
            LQ_WAIT_KEEP8();                                                                                          \
            __syncthreads();                                                                                          \
        }                                                                                                             \
    } while (0)
        if (t_fast > 0) {
            LQ_STEP(0, false, false);
            int t = 1;
            if constexpr (!TEXT && MODE == 2) {
                for (; t + 6 <= t_fast; t += 6) {
                    lp_window(t);
                    LQ_STEP_C(t, 0); LQ_STEP_C(t, 1); LQ_STEP_C(t, 2); LQ_STEP_C(t, 3); LQ_STEP_C(t, 4); LQ_STEP_C(t, 5);
                }
            }
            for (; t < t_fast; ++t) LQ_STEP(t, true, true);
            // drain: softmax of the last item, P.V of the last tile
            if constexpr (MODE == 2) {
                lq_bb<T, TEXT, 0, true, false, true>(sa, sb, nullptr, vslot(t_fast - 1), sXa, sXb, sYa, sYb, pXa, pXb, pYa,
                                                     pYb, k_addr, v_addr, P.qk_scale, frk);
                lq_bb<T, TEXT, 1, true, false, false>(sa, sb, nullptr, vslot(t_fast - 1), sYa, sYb, sXa, sXb, pYa, pYb, pXa,
                                                      pXb, k_addr, v_addr, P.qk_scale, frk);
            } else {
                lp_bb<T, TEXT, 0, true, false, true>(s1, nullptr, vslot(t_fast - 1), sXa, sYa, pXa, pYa, k_addr, v_addr,
                                                     P.qk_scale, frk);
                lp_bb<T, TEXT, 1, true, false, false>(s1, nullptr, vslot(t_fast - 1), sYa, sXa, pYa, pXa, k_addr, v_addr,
                                                      P.qk_scale, frk);
            }
        }
#undef LQ_STEP
#undef LQ_STEP_C
        if (!TEXT) {
            for (int t = t_fast; t < t_all; ++t) {
                issue_v(t);
                issue_k(t + 2);
                LP_WAIT_ALL();   // this step reads V^T(t) itself
                __syncthreads();
                const int blk = blk_at(t >> 1);
                if (MODE != 1)
                    lp_slow_tile<T>(sa, kslot(t), vslot(t), blk * 128 + (t & 1) * 64, blk >= P.text_block_start,
                                    P.text_amp, seqlen, hi, k_addr, v_addr);
                if (MODE != 0)
                    lp_slow_tile<T>(sb, kslot(t), vslot(t), blk * 128 + (t & 1) * 64, blk >= P.text_block_start,
                                    P.text_amp, seqlen, hi, k_addr, v_addr);
                __syncthreads();
            }
        }
        LP_WAIT_ALL();
        __syncthreads();      // the next walk's prologue overwrites ring slots other waves may still be reading
    };

    if (TEXT) {
        if (has_b) walk(LqMode<2>{}, nullptr, P.n_blocks);
        else walk(LqMode<0>{}, nullptr, P.n_blocks);
    } else {
        walk(LqMode<0>{}, list + n_sh, n_a);
        walk(LqMode<1>{}, list + n_sh + n_a, n_b);
        walk(LqMode<2>{}, list, n_sh);
    }

    // ---- epilogue: o = acc / l, rows >= seqlen written as zeros (image rows only) ----
    auto store = [&](const LpState& st, long long qrow) {
        uint16_t* const op = P.o + b * P.o_sb + qrow * P.o_ss + h * P.o_sh + hi * 4;
        const bool row_ok = TEXT || (qrow < seqlen);
#pragma unroll
        for (int db = 0; db < 4; ++db) {
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                uint2 w = make_uint2(0u, 0u);
                if (row_ok) {
                    w.x = pack2<T>(__fdiv_rn(st.o[db][rq * 4 + 0], st.l), __fdiv_rn(st.o[db][rq * 4 + 1], st.l));
                    w.y = pack2<T>(__fdiv_rn(st.o[db][rq * 4 + 2], st.l), __fdiv_rn(st.o[db][rq * 4 + 3], st.l));
                }
                *reinterpret_cast<uint2*>(op + db * 32 + rq * 8) = w;
            }
        }
    };
    store(sa, qrowA);
    if (has_b) store(sb, qrowB);
}

#define LQ_THREADS 256
constexpr int LQ_LDS_BYTES = 6 * LP_TILE;   // K ring 3 + V^T ring 3 tiles of 64 keys
// VARIANT 0: static mapping of query-block pairs to workgroups; 4: pairs drawn from per-XCD queues (lp_balance.h)
template <typename T, int VARIANT>
__global__ void __launch_bounds__(LQ_THREADS, 1) bsattn_lq_kernel(LqParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int n_text = P.n_blocks - P.nq_img;
    const int id = blockIdx.x;
    if (id < P.n_text_wg_pad) {   // text query blocks first: the longest work items start earliest
        if (id >= P.B * P.H * P.npair_txt) return;
        const int pr = id % P.npair_txt;
        const int bh = id / P.npair_txt;
        attn_pair_lq<T, true>(P, smem, bh / P.H, bh % P.H, P.nq_img + 2 * pr, 2 * pr + 1 < n_text, nullptr, 0, 0, 0);
        return;
    }
    int li = id - P.n_text_wg_pad;
    if (VARIANT == 4) {
        const int ticket = lp_draw_ticket(g_balance_ctr[P.bal_set], P.B * P.H, P.npair_img, P.xcd_chunk, li & 7,
                                          reinterpret_cast<int*>(smem));
        if (ticket < 0) return;
        const int y = ticket >> 28, t = ticket & 0x0fffffff;
        int nv = P.npair_img - y * P.xcd_chunk;
        nv = nv < P.xcd_chunk ? nv : P.xcd_chunk;
        li = __builtin_amdgcn_readfirstlane((t / nv) * P.img_per_head + (((t % nv) << 3) | y));
    }
    const int bh = li / P.img_per_head;
    const int r = li % P.img_per_head;
    int pr;
    if (P.xcd_chunk) {   // workgroup id -> XCD is id % 8: give each XCD a contiguous range of query-block pairs
        pr = (r & 7) * P.xcd_chunk + (r >> 3);
        if ((r >> 3) >= P.xcd_chunk || pr >= P.npair_img) return;
    } else {
        pr = r;
    }
    if (P.order) pr = P.order[(long long)bh * P.npair_img + pr];   // work-aware order inside the XCD's range
    const long long row = (long long)bh * P.npair_img + pr;
    const int32_t* list = P.pidx + row * P.n_blocks;
    const int n_sh = __builtin_amdgcn_readfirstlane(P.pcnt[row * 4 + 0]);
    const int n_a = __builtin_amdgcn_readfirstlane(P.pcnt[row * 4 + 1]);
    const int n_b = __builtin_amdgcn_readfirstlane(P.pcnt[row * 4 + 2]);
    attn_pair_lq<T, false>(P, smem, bh / P.H, bh % P.H, 2 * pr, 2 * pr + 1 < P.nq_img, list, n_sh, n_a, n_b);
}

template <typename T, int VARIANT>
static hipError_t lq_launch(const LqParams& P, long long grid, hipStream_t stream) {
    static bool smem_set[64] = {};
    lp_set_smem_once((const void*)bsattn_lq_kernel<T, VARIANT>, LQ_LDS_BYTES, smem_set);
    hipLaunchKernelGGL((bsattn_lq_kernel<T, VARIANT>), dim3((unsigned)grid), dim3(LQ_THREADS), LQ_LDS_BYTES, stream, P);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ list merge
// One wave per query-block pair: the two ascending kept lists -> [shared | A-only | B-only], each ascending.
constexpr int MERGE_WORDS = 128;   // up to 4096 kv blocks
__global__ void __launch_bounds__(64) pair_merge_kernel(const int32_t* __restrict__ idx, const int32_t* __restrict__ cnt,
                                                        int32_t* __restrict__ pidx, int32_t* __restrict__ pcnt,
                                                        int nq, int npair, int n_blocks) {
    __shared__ uint32_t bmA[MERGE_WORDS], bmB[MERGE_WORDS];
    const int lane = threadIdx.x;
    const long long prow = blockIdx.x;               // bh * npair + pr
    const long long bh = prow / npair;
    const int pr = (int)(prow % npair);
    const long long rowA = bh * nq + 2 * pr;
    const bool has_b = 2 * pr + 1 < nq;
    for (int w = lane; w < MERGE_WORDS; w += 64) bmA[w] = bmB[w] = 0u;
    __syncthreads();
    const int cA = cnt[rowA], cB = has_b ? cnt[rowA + 1] : 0;
    for (int i = lane; i < cA; i += 64) {
        const int j = idx[rowA * n_blocks + i];
        atomicOr(&bmA[j >> 5], 1u << (j & 31));
    }
    for (int i = lane; i < cB; i += 64) {
        const int j = idx[(rowA + 1) * n_blocks + i];
        atomicOr(&bmB[j >> 5], 1u << (j & 31));
    }
    __syncthreads();
    const int nwords = (n_blocks + 31) >> 5;
    int32_t* out = pidx + prow * n_blocks;
    int base = 0;
    for (int cat = 0; cat < 3; ++cat) {
        int run = 0;   // entries of this category written so far (wave-uniform)
        for (int w0 = 0; w0 < nwords; w0 += 64) {
            const int w = w0 + lane;
            uint32_t m = 0u;
            if (w < nwords) {
                const uint32_t a = bmA[w], bb = bmB[w];
                m = cat == 0 ? (a & bb) : cat == 1 ? (a & ~bb) : (bb & ~a);
            }
            const int c = __popc(m);
            int inc = c;   // inclusive wave scan
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(inc, o);
                if (lane >= o) inc += t;
            }
            int pos = base + run + inc - c;
            while (m) {
                const int bit = __ffs(m) - 1;
                m &= m - 1;
                out[pos++] = w * 32 + bit;
            }
            run += __shfl(inc, 63);
        }
        if (lane == 0) pcnt[prow * 4 + cat] = run;
        base += run;
    }
    if (lane == 0) pcnt[prow * 4 + 3] = 0;
}

}  // namespace
}  // namespace jenga

using namespace jenga;

extern "C" int jenga_pair_merge(void* stream, const int32_t* idx, const int32_t* cnt, int64_t B, int64_t H,
                                int64_t nq_img, int64_t n_blocks, int32_t* pidx, int32_t* pcnt) {
    if (B <= 0 || H <= 0 || nq_img < 0 || n_blocks <= 0 || (nq_img > 0 && (!idx || !cnt || !pidx || !pcnt))) {
        set_error("jenga_pair_merge: bad arguments");
        return JENGA_EINVAL;
    }
    if (n_blocks > MERGE_WORDS * 32) {
        set_error("jenga_pair_merge: at most %d kv blocks supported (got %lld)", MERGE_WORDS * 32, (long long)n_blocks);
        return JENGA_EUNSUPPORTED;
    }
    const long long npair = (nq_img + 1) / 2;
    const long long rows = B * H * npair;
    if (rows == 0) return JENGA_OK;
    hipLaunchKernelGGL(pair_merge_kernel, dim3((unsigned)rows), dim3(64), 0, (hipStream_t)stream, idx, cnt, pidx, pcnt,
                       (int)nq_img, (int)npair, (int)n_blocks);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("jenga_pair_merge: %s", hipGetErrorString(e));
        return JENGA_ELAUNCH;
    }
    return JENGA_OK;
}

extern "C" int jenga_bsattn_pair_fwd(void* stream, const void* q, const void* k, const void* vt, void* o,
                                     const int32_t* seqlens, const int32_t* pidx, const int32_t* pcnt,
                                     const int32_t* order, int64_t B, int64_t H, int64_t n_blocks, int64_t nq_img, int64_t q_sb, int64_t q_ss,
                                     int64_t q_sh, int64_t k_sb, int64_t k_ss, int64_t k_sh, int64_t o_sb, int64_t o_ss,
                                     int64_t o_sh, float sm_scale, float text_amp, int64_t text_block_start, int dtype,
                                     int flags) {
    if (!q || !k || !vt || !o || B <= 0 || H <= 0 || n_blocks <= 0 || nq_img < 0 || nq_img > n_blocks) {
        set_error("jenga_bsattn_pair_fwd: bad arguments");
        return JENGA_EINVAL;
    }
    if (nq_img > 0 && (!pidx || !pcnt)) {
        set_error("jenga_bsattn_pair_fwd: pidx/pcnt are required when nq_img > 0");
        return JENGA_EINVAL;
    }
    const int64_t strides[9] = {q_sb, q_ss, q_sh, k_sb, k_ss, k_sh, o_sb, o_ss, o_sh};
    for (int i = 0; i < 9; ++i)
        if (strides[i] & 7) {
            set_error("jenga_bsattn_pair_fwd: strides must be multiples of 8 elements (16-byte rows)");
            return JENGA_EINVAL;
        }
    if (((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)vt & 15) || ((uintptr_t)o & 15)) {
        set_error("jenga_bsattn_pair_fwd: pointers must be 16-byte aligned");
        return JENGA_EINVAL;
    }
    if (dtype != JENGA_BF16 && dtype != JENGA_FP16) {
        set_error("jenga_bsattn_pair_fwd: dtype must be bf16 or fp16");
        return JENGA_EUNSUPPORTED;
    }
    if (k_ss < 0 || k_ss >= (1LL << 31)) {
        set_error("jenga_bsattn_pair_fwd: k token stride %lld out of range", (long long)k_ss);
        return JENGA_EINVAL;
    }
    LqParams P;
    P.q = (const uint16_t*)q;
    P.k = (const uint16_t*)k;
    P.vt = (const uint16_t*)vt;
    P.o = (uint16_t*)o;
    P.seqlens = seqlens;
    P.pidx = pidx;
    P.pcnt = pcnt;
    P.order = order;
    P.q_sb = q_sb; P.q_ss = q_ss; P.q_sh = q_sh;
    P.k_sb = k_sb; P.k_ss = k_ss; P.k_sh = k_sh;
    P.o_sb = o_sb; P.o_ss = o_ss; P.o_sh = o_sh;
    P.B = (int)B; P.H = (int)H; P.n_blocks = (int)n_blocks; P.nq_img = (int)nq_img;
    P.npair_img = (int)((nq_img + 1) / 2);
    const long long n_text = n_blocks - nq_img;
    P.npair_txt = (int)((n_text + 1) / 2);
    P.text_block_start = (int)text_block_start;
    P.qk_scale = (float)((double)sm_scale * 1.44269504);
    P.text_amp = text_amp;
    P.bal_set = 0;
    const long long n_text_wg = B * H * (long long)P.npair_txt;
    P.n_text_wg_pad = (int)((n_text_wg + 7) / 8 * 8);
    if ((flags & JENGA_ATTN_XCD_REMAP) && P.npair_img >= 64) {
        P.xcd_chunk = (P.npair_img + 7) / 8;
        P.img_per_head = P.xcd_chunk * 8;
    } else {
        P.xcd_chunk = 0;
        P.img_per_head = P.npair_img;
    }
    const long long grid = (long long)P.n_text_wg_pad + B * H * (long long)P.img_per_head;
    if (grid <= 0 || grid > 0x7fffffffLL) {
        set_error("jenga_bsattn_pair_fwd: grid size %lld out of range", grid);
        return JENGA_EINVAL;
    }
    if (flags & JENGA_ATTN_BALANCE) {      // (per-device state ordered by events: a capturing stream gets the static launch)
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing((hipStream_t)stream, &cap) != hipSuccess) {
            (void)hipGetLastError();
            cap = hipStreamCaptureStatusActive;
        }
        if (cap != hipStreamCaptureStatusNone) flags &= ~JENGA_ATTN_BALANCE;
    }
    long long grid_bal = 0;
    int bal_slot = -1;
    if ((flags & JENGA_ATTN_BALANCE) && P.xcd_chunk && (P.n_text_wg_pad & 7) == 0 && (P.img_per_head & 7) == 0) {
        const long long img = B * H * (long long)P.img_per_head;
        long long extra = ((img * lp_balance_extra_pct() / 100) + 7) & ~7LL;
        extra = extra < 8 ? 8 : extra;
        if (grid + extra <= 0x7fffffffLL && B * H * (long long)P.xcd_chunk < (1LL << 28)) {
            bal_slot = lp_balance_acquire((hipStream_t)stream);
            if (bal_slot >= 0) {
                grid_bal = grid + extra;
                P.bal_set = bal_slot;
            }
        }
    }
    hipError_t e;
    if (grid_bal)
        e = dtype == JENGA_BF16 ? lq_launch<BF16, 4>(P, grid_bal, (hipStream_t)stream)
                                : lq_launch<FP16, 4>(P, grid_bal, (hipStream_t)stream);
    else
        e = dtype == JENGA_BF16 ? lq_launch<BF16, 0>(P, grid, (hipStream_t)stream)
                                : lq_launch<FP16, 0>(P, grid, (hipStream_t)stream);
    if (bal_slot >= 0) lp_balance_release(bal_slot, (hipStream_t)stream);
    if (e != hipSuccess) {
        set_error("jenga_bsattn_pair_fwd: %s", hipGetErrorString(e));
        return JENGA_ELAUNCH;
    }
    return JENGA_OK;
}
