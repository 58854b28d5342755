// Shared device code of the LP attention kernels (bsattn3.hip: one query block per 4-wave workgroup, two workgroups per CU;
// bsattn5.hip: a PAIR of query blocks per 4-wave workgroup, one workgroup per CU): LDS tile geometry, LDS-DMA helpers, the per-wave softmax
// state, the exact (max-first) path, the 16-slot pipelined basic block lp_bb and the unpipelined tail tile.
// See the header of bsattn3.hip for the design.
#pragma once
#include "common.h"

namespace jenga {
namespace {

constexpr int LP_TILE = 16384;
constexpr int LP_K_RING = 0;               // 3 slots: tile t in slot t % 3
constexpr int LP_V_RING = 3 * LP_TILE;     // 2 slots: tile t in slot t & 1
constexpr int LP_LDS_BYTES = 5 * LP_TILE;  // 80 KiB: two workgroups per CU

constexpr float LP_RAISE_SUM = 256.0f;
template <typename T> __device__ __forceinline__ constexpr float lp_tiny();
template <> __device__ __forceinline__ constexpr float lp_tiny<BF16>() { return 8.673617379884035e-19f; }   // 2^-60
template <> __device__ __forceinline__ constexpr float lp_tiny<FP16>() { return 0.0625f; }                   // 2^-4

// ---- round 6: the "issue-slot diet" switches (review item 1; both kernels can be built with either, A/B in profiles/r06_*) ----
// LP_ROWSUM_DOT2: the row sums l come from the PACKED 16-bit P with one v_dot2_f32_{bf16,f16} per register pair (P . (1, 1) + acc)
//   instead of one fp32 add per score: 8 instead of 16 + 3 VALU instructions per 32-key item and sub-block.  Deviation from the
//   reference, which sums the UNROUNDED fp32 p (attention_block_triton_diffres.py:131 `l_i = l_i * alpha + tl.sum(p, 1)`): here
//   l = sum of dtype-rounded p, i.e. numerator and denominator of o = acc / l see the same P.  |dl / l| <= 2^-9 / sqrt(n) for
//   bf16; the goldens' <= 2 ulp bound is re-checked with it (tests/test_gpu_parity.py).
#ifndef LP_ROWSUM_DOT2
#define LP_ROWSUM_DOT2 0
#endif
// LP_DMA_M0_ONCE: one M0 write per four-piece LDS-DMA stage in the unrolled loops (lp_stage1<0> writes it, <1..3> rely on it):
//   3 x (s_mov + s_nop) fewer per stage.  M0 is per-wave state; nothing else in these kernels writes it between the pieces of a
//   stage -- tools/isa_hazards.py checks that in the generated code (no M0 write between an lp_stage1<0> and the third
//   global_load_lds behind it, all four in one basic block).
#ifndef LP_DMA_M0_ONCE
#define LP_DMA_M0_ONCE 0
#endif

typedef __bf16 lp_bf2 __attribute__((ext_vector_type(2)));
typedef _Float16 lp_h2 __attribute__((ext_vector_type(2)));
// c + lo(w) + hi(w) for a packed pair of T
template <typename T>
__device__ __forceinline__ float lp_pair_sum(uint32_t w, float c) {
    if constexpr (__is_same(T, BF16))
        return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(lp_bf2, w), __builtin_bit_cast(lp_bf2, 0x3f803f80u), c, false);
    else
        return __builtin_amdgcn_fdot2(__builtin_bit_cast(lp_h2, w), __builtin_bit_cast(lp_h2, 0x3c003c00u), c, false);
}

// four 1-KiB LDS-DMA pieces of one tile (the instruction's immediate offset adds to BOTH addresses: piece i's lane offsets
// are biased by -1024 i, so one M0 value serves the four pieces)
__device__ __forceinline__ void lp_stage4(const void* base, unsigned lds, unsigned o0, unsigned o1, unsigned o2,
                                          unsigned o3) {
    asm volatile("s_mov_b32 m0, %0\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %2, %1\n\t"
                 "global_load_lds_dwordx4 %3, %1 offset:1024\n\t"
                 "global_load_lds_dwordx4 %4, %1 offset:2048\n\t"
                 "global_load_lds_dwordx4 %5, %1 offset:3072"
                 :
                 : "s"(lds), "s"(base), "v"(o0), "v"(o1), "v"(o2), "v"(o3)
                 : "memory", "m0");   // M0 = LDS base of the DMA: the compiler must not assume it survives
}
// one piece: in the unrolled main loop the four pieces of a stage go out in four MFMA slots of the block (1, 5, 9,
// 13) instead of back to back in front of it -- less queueing in the vector-memory path, +2 % sustained
struct LpDma {
    const void* base;
    unsigned lds;
    unsigned o[4];
};
template <int I>
__device__ __forceinline__ void lp_stage1(const LpDma& d) {
    if (I == 0)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1" : : "s"(d.lds), "s"(d.base), "v"(d.o[0]) : "memory", "m0");
#if LP_DMA_M0_ONCE
    else if (I == 1)
        asm volatile("global_load_lds_dwordx4 %1, %0 offset:1024" : : "s"(d.base), "v"(d.o[1]) : "memory");
    else if (I == 2)
        asm volatile("global_load_lds_dwordx4 %1, %0 offset:2048" : : "s"(d.base), "v"(d.o[2]) : "memory");
    else
        asm volatile("global_load_lds_dwordx4 %1, %0 offset:3072" : : "s"(d.base), "v"(d.o[3]) : "memory");
#else
    else if (I == 1)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1 offset:1024" : : "s"(d.lds), "s"(d.base), "v"(d.o[1]) : "memory", "m0");
    else if (I == 2)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1 offset:2048" : : "s"(d.lds), "s"(d.base), "v"(d.o[2]) : "memory", "m0");
    else
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1 offset:3072" : : "s"(d.lds), "s"(d.base), "v"(d.o[3]) : "memory", "m0");
#endif
}
#define LP_WAIT_ALL() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define LP_WAIT_KEEP4() asm volatile("s_waitcnt vmcnt(4)" ::: "memory")

// QK^T MFMAs.  Default: the compiler's builtin.  With LP_QK_MFMA_ASM (the pair kernel, bsattn5.hip: one wave per SIMD, 512
// registers) the instruction is written out with explicit register classes -- D / C (the scores) in architectural VGPRs, the
// Q operand in the accumulator half of the register file: at that budget hipcc selects the "AGPR form" for every builtin
// MFMA (scores land in AGPRs, one v_accvgpr_read per score in front of the softmax), rematerialises the -m~ C operand with 16
// v_accvgpr_write per item and parks Q in AGPRs only to copy it back in front of every MFMA (4 v_accvgpr_read each).
// hipcc's hazard recogniser does not look into asm statements; what covers each hazard is listed at lq_bb (bsattn5.hip).
#ifndef LP_QK_MFMA_ASM
#define LP_QK_MFMA_ASM 0
#endif
typedef unsigned lp_u32x4 __attribute__((ext_vector_type(4)));
// a Q fragment (8 x 16 bit per lane).  asm form: a register TUPLE type that is handed to the asm statements as it is -- built
// from a uint4 per use, its four dwords live wherever the allocator put them and are gathered into a fresh AGPR quad in front
// of every MFMA
#if LP_QK_MFMA_ASM
typedef lp_u32x4 LpQ;
#else
typedef uint4 LpQ;
#endif
// d = a . b + c  (c another register tuple: the first MFMA of an item)
template <typename T>
__device__ __forceinline__ void lp_qk_first(f32x16& d, const uint4& a, const LpQ& b, const f32x16& c) {
#if LP_QK_MFMA_ASM
    const lp_u32x4 a4 = __builtin_bit_cast(lp_u32x4, a);
    if constexpr (__is_same(T, BF16))
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a4), "a"(b), "v"(c));
    else
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(d) : "v"(a4), "a"(b), "v"(c));
#else
    d = mfma32<T>(a, b, c);
#endif
}
// d = a . b  (TEXT rows: no C operand)
template <typename T>
__device__ __forceinline__ void lp_qk_zero(f32x16& d, const uint4& a, const LpQ& b, const f32x16& zero16) {
#if LP_QK_MFMA_ASM
    (void)zero16;
    const lp_u32x4 a4 = __builtin_bit_cast(lp_u32x4, a);
    if constexpr (__is_same(T, BF16))
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(a4), "a"(b));
    else
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(d) : "v"(a4), "a"(b));
#else
    d = mfma32<T>(a, b, zero16);
#endif
}
// d += a . b
template <typename T>
__device__ __forceinline__ void lp_qk_acc(f32x16& d, const uint4& a, const LpQ& b) {
#if LP_QK_MFMA_ASM
    const lp_u32x4 a4 = __builtin_bit_cast(lp_u32x4, a);
    if constexpr (__is_same(T, BF16))
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a4), "a"(b));
    else
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a4), "a"(b));
#else
    d = mfma32<T>(a, b, d);
#endif
}
// (asm form) the scores of an item were written by MFMAs the compiler cannot see: before VALU code reads them with fewer
// than two MFMA issue times in between (blocks without P.V MFMAs behind the QK^T ones)
__device__ __forceinline__ void lp_qk_settle() {
#if LP_QK_MFMA_ASM
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#endif
}
// (asm form) behind VALU writes to a C operand (the exact path's new -m~) and in front of the next MFMA reading it
__device__ __forceinline__ void lp_qk_c_written() {
#if LP_QK_MFMA_ASM
    asm volatile("s_nop 3" ::: "memory");
#endif
}
// (asm form) keep the 16 copies of -m~ a register tuple of their own: seen as 16 equal values the tuple is rebuilt with 16
// moves in front of every item
__device__ __forceinline__ void lp_cinit_pin(f32x16& c) {
#if LP_QK_MFMA_ASM
    asm volatile("" : "+v"(c));
#else
    (void)c;
#endif
}

struct LpState {
    LpQ qf[8];
    f32x16 o[4];
    float l;       // whole-row running sum (identical in the two lanes that share a row)
    float neg_m;   // -m~
    f32x16 cinit;  // 16 copies of -m~: C operand of the first QK^T MFMA of an item (image rows), so that the scores
                   // arrive as S - m~ and the softmax saves one VALU instruction per score
};

// exact (max-first) softmax of one 32-key item from its intact scores
// `s`: raw scores (TEXT) or S - m~(old) (image rows).  `pend`: scores of the NEXT item, already produced against the
// old m~ (or null).
template <typename T, bool TEXT>
__device__ __forceinline__ void lp_exact(LpState& st, const f32x16& s, uint4 (&pf)[2], float& psum, float qk_scale,
                                         f32x16* pend) {
#if LP_QK_MFMA_ASM
    // Same arithmetic as below, written for the 512-register kernel: (a) no value arrays (16 + 16 temporaries here pushed
    // long-lived tuples -- the -m~ C operands -- out of the architectural VGPRs for the WHOLE main loop); (b) O lives in
    // the accumulator registers and is rescaled IN PLACE there: written as plain `o *= f2`, hipcc keeps a rescaled copy in
    // VGPRs and pays for it on the hot path (v_accvgpr moves at the join).
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, TEXT ? s[r] * qk_scale + st.neg_m : s[r]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const bool move = (tmax > 0.f) || (st.l + psum < lp_tiny<T>());
    const float delta = (move && tmax > -1e30f) ? fmaxf(ceilf(tmax), -120.f) : 0.f;
    const float f2 = __builtin_amdgcn_exp2f(-delta);
    const float old_neg_m = st.neg_m;
    st.neg_m -= delta;
    st.l *= f2;
    // s_nop: the last P.V MFMA of an accumulator may still be in flight, and nothing inside an asm statement is covered by
    // the compiler's hazard recogniser; the s_nop behind the last write covers v_accvgpr_write -> MFMA srcC
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float t_;
            asm volatile("v_accvgpr_read_b32 %1, %0\n\tv_mul_f32 %1, %1, %2\n\ts_nop 0\n\tv_accvgpr_write_b32 %0, %1"
                         : "+a"(st.o[i][r]), "=&v"(t_)
                         : "v"(f2));
        }
    asm volatile("s_nop 3" ::: "memory");
    if (!TEXT) {
#pragma unroll
        for (int r = 0; r < 16; ++r) st.cinit[r] = st.neg_m;
        lp_cinit_pin(st.cinit);
        if (pend) {
#pragma unroll
            for (int r = 0; r < 16; ++r) (*pend)[r] -= delta;
        }
    }
    psum = 0.f;
    uint32_t w[8];
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const float e0 = __builtin_amdgcn_exp2f((TEXT ? s[r] * qk_scale + old_neg_m : s[r]) - delta);
        const float e1 = __builtin_amdgcn_exp2f((TEXT ? s[r + 1] * qk_scale + old_neg_m : s[r + 1]) - delta);
        w[r >> 1] = pack2<T>(e0, e1);
#if LP_ROWSUM_DOT2
        psum = lp_pair_sum<T>(w[r >> 1], psum);
#else
        psum += e0;
        psum += e1;
#endif
    }
    psum += __shfl_xor(psum, 32);
    pf[0] = make_uint4(w[0], w[1], w[2], w[3]);
    pf[1] = make_uint4(w[4], w[5], w[6], w[7]);
    lp_qk_c_written();
#else
    float v[16];
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        v[r] = TEXT ? s[r] * qk_scale + st.neg_m : s[r];
        tmax = fmaxf(tmax, v[r]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const bool move = (tmax > 0.f) || (st.l + psum < lp_tiny<T>());
    const float delta = (move && tmax > -1e30f) ? fmaxf(ceilf(tmax), -120.f) : 0.f;
    const float f2 = __builtin_amdgcn_exp2f(-delta);
    st.neg_m -= delta;
    st.l *= f2;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) st.o[i][r] *= f2;
    if (!TEXT) {
#pragma unroll
        for (int r = 0; r < 16; ++r) st.cinit[r] = st.neg_m;
        lp_cinit_pin(st.cinit);
        if (pend) {
#pragma unroll
            for (int r = 0; r < 16; ++r) (*pend)[r] -= delta;
        }
    }
    float e[16];
    psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        e[r] = __builtin_amdgcn_exp2f(v[r] - delta);
#if !LP_ROWSUM_DOT2
        psum += e[r];
#endif
    }
    pf[0] = make_uint4(pack2<T>(e[0], e[1]), pack2<T>(e[2], e[3]), pack2<T>(e[4], e[5]), pack2<T>(e[6], e[7]));
    pf[1] = make_uint4(pack2<T>(e[8], e[9]), pack2<T>(e[10], e[11]), pack2<T>(e[12], e[13]), pack2<T>(e[14], e[15]));
#if LP_ROWSUM_DOT2
    psum = lp_pair_sum<T>(pf[0].x, psum); psum = lp_pair_sum<T>(pf[0].y, psum);
    psum = lp_pair_sum<T>(pf[0].z, psum); psum = lp_pair_sum<T>(pf[0].w, psum);
    psum = lp_pair_sum<T>(pf[1].x, psum); psum = lp_pair_sum<T>(pf[1].y, psum);
    psum = lp_pair_sum<T>(pf[1].z, psum); psum = lp_pair_sum<T>(pf[1].w, psum);
#endif
    psum += __shfl_xor(psum, 32);
#endif
}

// One basic block of the pipeline (see the header).  HALF: which 32-key half of the 64-key tiles kt (item i) and vt
// (item i-2) this block works on.
// KOFF / VOFF: byte offset of the ring slot inside the K / V^T ring when it is known at compile time (the unrolled
// main loop) -- it then folds into the ds_read offset field together with the HALF / d-block offsets and the eight K
// + two V^T per-item address adds disappear; -1: the slot is in the pointer (kt / vt = smem + slot * LP_TILE).
// PRE: 1 = during its P.V MFMAs this (HALF 0) block also issues the K fragment reads of the NEXT block (HALF 1 of the
// same tile, into frk, which lives in the caller), 2 = this block's K fragments were issued that way (no up-front
// reads): the fragment pipeline then runs through the block boundary instead of draining and refilling there.
template <typename T, bool TEXT, int HALF, bool DO_PV, bool DO_QK, bool DO_SM, int KOFF = -1, int VOFF = -1, int PRE = 0>
__device__ __forceinline__ void lp_bb(LpState& st, const unsigned char* kt, const unsigned char* vt, f32x16& sn,
                                      const f32x16& sp, const uint4 (&pf_old)[2], uint4 (&pf_new)[2],
                                      const int (&k_addr)[8], const int (&v_addr)[4], float qk_scale,
                                      uint4 (&frk)[8], const LpDma* dma = nullptr) {
    constexpr int KO = KOFF < 0 ? 0 : KOFF, VO = VOFF < 0 ? 0 : VOFF;
    static_assert(PRE == 0 || (DO_QK && DO_PV), "the cross-block fragment pipeline is for full blocks");
    static_assert(PRE != 1 || HALF == 0, "only the first half prefetches (the next tile may still be in flight)");
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
    uint4 frv[8];
    float tt[16], xx[16];
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    float half_ = 0.f;
    uint32_t ww[8];
#define LP_READ(F_)                                                                                                   \
    do {                                                                                                              \
        if ((F_) < 8) {                                                                                               \
            if (DO_QK) frk[(F_) & 7] = *reinterpret_cast<const uint4*>(kt + k_addr[(F_) & 7] + (HALF * 8192 + KO));   \
        } else if ((F_) < 16) {                                                                                       \
            if (DO_PV) frv[(F_) & 7] = *reinterpret_cast<const uint4*>(vt + v_addr[2 * HALF + (((F_) - 8) >> 2)] +    \
                                                                       ((((F_) - 8) & 3) * 4096 + VO));               \
        } else if ((F_) < 24 && PRE == 1) { /* K fragment (F_ - 16) of the next block: the other half of this tile */ \
            frk[(F_) & 7] = *reinterpret_cast<const uint4*>(kt + k_addr[(F_) & 7] + (8192 + KO));                     \
        }                                                                                                             \
    } while (0)
    /* image rows: the scores are S - m~ already (C operand); TEXT rows: raw scores, scaled and shifted here.
       Image rows: all 16 scores exist when the block starts, so element e is exponentiated in slot X(e) = e / 2 for
       e < 4, e - 2 after that, added / packed one slot later, and the 4-way sum tree closes in slot 15: what is left
       behind the last MFMA is the cross-half permlane and the ballot.  TEXT rows keep the three-stage form (their
       scale-and-shift stage needs a slot of its own) that ends two slots later. */
#define LP_SM_X(E_) ((E_) < 4 ? ((E_) >> 1) : (E_) - 2)
#define LP_SM(M_)                                                                                                     \
    do {                                                                                                              \
        if (DO_SM) {                                                                                      \
            if (TEXT) {                                                                                               \
                if ((M_) < 16) tt[(M_) & 15] = sp[(M_) & 15] * qk_scale + st.neg_m;                                   \
                if ((M_) >= 1 && (M_) < 17) xx[((M_) - 1) & 15] = __builtin_amdgcn_exp2f(tt[((M_) - 1) & 15]);        \
                if ((M_) >= 2 && (M_) < 18) {                                                                         \
                    if (!LP_ROWSUM_DOT2) {                                                                            \
                        if ((M_) < 6) acc[((M_) - 2) & 3] = xx[((M_) - 2) & 15];                                      \
                        else acc[((M_) - 2) & 3] += xx[((M_) - 2) & 15];                                              \
                    }                                                                                                 \
                    if (((M_) - 2) & 1) {                                                                             \
                        ww[(((M_) - 2) & 15) >> 1] = pack2<T>(xx[((M_) - 3) & 15], xx[((M_) - 2) & 15]);              \
                        /* (the dot2 in front of the pin: hipcc's hazard recogniser assumes a dst_sel forwarding hazard */ \
                        /* behind every inline asm that defines a VGPR and puts an s_nop in front of an adjacent reader) */ \
                        if (LP_ROWSUM_DOT2) acc[0] = lp_pair_sum<T>(ww[(((M_) - 2) & 15) >> 1], (M_) == 3 ? 0.f : acc[0]); \
                        asm volatile("" : "+v"(ww[(((M_) - 2) & 15) >> 1]));   /* stay in this slot */                \
                    }                                                                                                 \
                }                                                                                                     \
                if ((M_) == 17) half_ = LP_ROWSUM_DOT2 ? acc[0] : (acc[0] + acc[1]) + (acc[2] + acc[3]);              \
            } else {                                                                                                  \
                _Pragma("unroll") for (int e_ = 0; e_ < 16; ++e_) {                                                   \
                    if (LP_SM_X(e_) + 1 == (M_)) {                                                                    \
                        if (!LP_ROWSUM_DOT2) { if (e_ < 4) acc[e_ & 3] = xx[e_]; else acc[e_ & 3] += xx[e_]; }        \
                        if (e_ & 1) {                                                                                 \
                            ww[e_ >> 1] = pack2<T>(xx[e_ - 1], xx[e_]);                                               \
                            if (LP_ROWSUM_DOT2) acc[0] = lp_pair_sum<T>(ww[e_ >> 1], e_ == 1 ? 0.f : acc[0]);         \
                            asm volatile("" : "+v"(ww[e_ >> 1]));   /* stay in this slot */                           \
                        }                                                                                             \
                    }                                                                                                 \
                }                                                                                                     \
                _Pragma("unroll") for (int e_ = 0; e_ < 16; ++e_)                                                     \
                    if (LP_SM_X(e_) == (M_)) xx[e_] = __builtin_amdgcn_exp2f(sp[e_]);                                 \
                if ((M_) == 15) half_ = LP_ROWSUM_DOT2 ? acc[0] : (acc[0] + acc[1]) + (acc[2] + acc[3]);              \
            }                                                                                                         \
        }                                                                                                             \
    } while (0)
#define LP_MFMA(D_, A_, B_, C_) D_ = mfma32<T>(A_, B_, C_)
    /* fragment reads go out two at a time, eight MFMAs ahead (slot m, m even, reads the fragments of MFMAs m+8 and
       m+9), with ONE explicit counted s_waitcnt lgkmcnt per two MFMAs (the compiler would emit one per MFMA).
       Before MFMA m (even) fragments m, m+1 must be there; the reads issued behind them are m+2 .. min(m+7, 15). */
#define LP_LGKM(N_)                                                                                                   \
    do {                                                                                                              \
        if ((N_) == 0) __builtin_amdgcn_s_waitcnt(0xC07F);                                                            \
        else if ((N_) == 2) __builtin_amdgcn_s_waitcnt(0xC27F);                                                       \
        else if ((N_) == 4) __builtin_amdgcn_s_waitcnt(0xC47F);                                                       \
        else if ((N_) == 6) __builtin_amdgcn_s_waitcnt(0xC67F);                                                       \
        else if ((N_) == 8) __builtin_amdgcn_s_waitcnt(0xC87F);                                                       \
        else __builtin_amdgcn_s_waitcnt(0xC07F);                                                                      \
    } while (0)
#define LP_SLOT(M_)                                                                                                   \
    do {                                                                                                              \
        if (!((M_) & 1) && (((M_) < 8 && DO_QK) || ((M_) >= 8 && DO_PV))) {                                           \
            if (!DO_QK) LP_LGKM(14 - (M_) > 8 ? 0 : 14 - (M_));   /* drain forms: fragments 8..15 read up front */    \
            else if (PRE == 1) LP_LGKM(6);   /* the next block's reads keep the queue at eight */                     \
            else LP_LGKM((M_) <= 8 ? 6 : 14 - (M_));                                                                  \
            __builtin_amdgcn_sched_barrier(0);   /* or hipcc moves the MFMA above the wait and adds its own */        \
        }                                                                                                             \
        if ((M_) < 8) {                                                                                               \
            if (DO_QK) {                                                                                              \
                if ((M_) == 0) { if (TEXT) lp_qk_zero<T>(sn, frk[(M_) & 7], st.qf[(M_) & 7], zero16); else lp_qk_first<T>(sn, frk[(M_) & 7], st.qf[(M_) & 7], st.cinit); } \
                else lp_qk_acc<T>(sn, frk[(M_) & 7], st.qf[(M_) & 7]);                                                \
            }                                                                                                         \
        } else if (DO_PV) {                                                                                           \
            LP_MFMA(st.o[((M_) - 8) & 3], frv[(M_) & 7], pf_old[((M_) - 8) >> 2], st.o[((M_) - 8) & 3]);              \
        }                                                                                                             \
        if (((M_) & 1) == 0) {                                                                                        \
            LP_READ((M_) + 8); LP_READ((M_) + 9);                                                                     \
        }                                                                                                             \
        if (dma) {                                                                                                    \
            if ((M_) == 1) lp_stage1<0>(*dma);                                                                        \
            if ((M_) == 5) lp_stage1<1>(*dma);                                                                        \
            if ((M_) == 9) lp_stage1<2>(*dma);                                                                        \
            if ((M_) == 13) lp_stage1<3>(*dma);                                                                       \
        }                                                                                                             \
        LP_SM(M_);                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
    } while (0)
    if (PRE != 2) {
        LP_READ(0); LP_READ(1); LP_READ(2); LP_READ(3); LP_READ(4); LP_READ(5); LP_READ(6); LP_READ(7);
    }
    if (!DO_QK) {
        LP_READ(8); LP_READ(9); LP_READ(10); LP_READ(11); LP_READ(12); LP_READ(13); LP_READ(14); LP_READ(15);
    }
    __builtin_amdgcn_sched_barrier(0);
    LP_SLOT(0); LP_SLOT(1); LP_SLOT(2); LP_SLOT(3); LP_SLOT(4); LP_SLOT(5); LP_SLOT(6); LP_SLOT(7);
    LP_SLOT(8); LP_SLOT(9); LP_SLOT(10); LP_SLOT(11); LP_SLOT(12); LP_SLOT(13); LP_SLOT(14); LP_SLOT(15);
    LP_SM(16);
    LP_SM(17);
#undef LP_READ
#undef LP_SM
#undef LP_SM_X
#undef LP_SLOT
#undef LP_MFMA
#undef LP_LGKM
    if (DO_QK && !DO_PV) lp_qk_settle();
    if (DO_SM) {
        const auto sw_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(half_), __float_as_uint(half_), false, false);
        float psum = __uint_as_float(sw_[0]) + __uint_as_float(sw_[1]);   // both half-lanes: the row's 32 keys
        pf_new[0] = make_uint4(ww[0], ww[1], ww[2], ww[3]);
        pf_new[1] = make_uint4(ww[4], ww[5], ww[6], ww[7]);
        // two ballots straight off the compares (a ballot of an OR-ed condition goes through v_cndmask + v_cmp_ne)
        if ((__builtin_amdgcn_ballot_w64(!(psum <= LP_RAISE_SUM)) |
             __builtin_amdgcn_ballot_w64(st.l + psum < lp_tiny<T>())) != 0ull)
            lp_exact<T, TEXT>(st, sp, pf_new, psum, qk_scale, DO_QK ? &sn : nullptr);
        st.l += psum;
    }
}

// a whole 64-key tile, unpipelined, with the text_amp add and the kv-length mask (tail of the ascending lists)
// TEXT (dense mode: unscaled Q in the registers, the scale applied to the fp32 scores): the ragged last tile of a
// cross-attention kv sequence (jenga_cross_attn_fwd with kv_len not a multiple of 64)
template <typename T, bool TEXT = false>
__device__ __forceinline__ void lp_slow_tile(LpState& st, const unsigned char* kt, const unsigned char* vt, int key0,
                                             bool amp_on, float text_amp, int seqlen, int hi,
                                             const int (&k_addr)[8], const int (&v_addr)[4], float qk_scale = 0.f) {
    if (key0 >= seqlen) return;   // contributes exp2(-inf) = 0
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        f32x16 s;
        {
            uint4 ka[8];
#pragma unroll
            for (int ds = 0; ds < 8; ++ds) ka[ds] = *reinterpret_cast<const uint4*>(kt + k_addr[ds] + half * 8192);
            lp_qk_first<T>(s, ka[0], st.qf[0], st.cinit);   // S - m~
#pragma unroll
            for (int ds = 1; ds < 8; ++ds) lp_qk_acc<T>(s, ka[ds], st.qf[ds]);
            lp_qk_settle();
        }
        if (amp_on) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] += text_amp;
        }
        if (key0 + 64 > seqlen) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kk = key0 + half * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (kk >= seqlen) s[r] = -INFINITY;
            }
        }
        uint4 pf[2];
        float psum = 0.f;
        lp_exact<T, TEXT>(st, s, pf, psum, qk_scale, nullptr);
        st.l += psum;
        uint4 va[2][4];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int db = 0; db < 4; ++db)
                va[ks][db] = *reinterpret_cast<const uint4*>(vt + v_addr[2 * half + ks] + db * 4096);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int db = 0; db < 4; ++db) st.o[db] = mfma32<T>(va[ks][db], pf[ks], st.o[db]);
    }
}

}  // namespace
}  // namespace jenga
