// Block-sparse attention forward, third generation ("LP": the round-1 decomposition with an in-wave software
// pipeline), gfx950, head_dim 128, 128-token blocks.  Same arguments, lists and numerics as jenga_bsattn_fwd.
//
// What two measured kernels taught:
//   * bsattn.hip (round 1): 4 waves = one 128-row query block, two workgroups per CU = two waves per SIMD.  The
//     softmax VALU work of one wave does not hide under the OTHER wave's MFMAs on a gfx950 SIMD, so MFMA time and
//     VALU time add up: 1040 TFLOP/s.
//   * bsattn2.hip (pair kernel): one wave per SIMD, softmax of item i-1 placed instruction by instruction into the
//     MFMA gaps of items i and i-2 IN THE SAME WAVE -- that works: the compute-only build runs at 1410-1460 TFLOP/s.
//     But every 1-KiB LDS-DMA piece parks its issuing wave for ~130 cycles (L2-hot or not, wherever it is placed,
//     whether or not M0 is restored), and with one wave per SIMD nothing covers that: 935-1015 TFLOP/s.
// This kernel takes both halves: round 1's workgroup shape and LDS ring (so that the partner wave of the SIMD -- a
// wave of the CU's other workgroup -- runs while this one sits in its DMA issue), and the pair kernel's in-wave
// pipeline, cut to 32-key half tiles so that it fits the 256 registers a wave has at two waves per SIMD:
//   item     = (this wave's 32 query rows) x (32 keys): 8 QK^T MFMAs -> softmax of 16 scores/lane -> 8 P.V MFMAs
//   block i  : MFMA 0-7  S(i)    = K . Q~^T                    \   16 fenced slots: MFMA | fragment read 8 ahead |
//              MFMA 8-15 O      += V^T(i-2) . P(i-2)            >  one score of item i-1 in three skewed stages
//              VALU      P(i-1) = bf16(exp2(S(i-1) - m~)), l   /   (t = s - m~ | exp2 | row-sum add + bf16 pack)
//   then the wave-uniform ballot for the exact max-first path of item i-1 (its P.V has not started, the P.V of item
//   i-2 is complete: guide T13's safe order).
// Tile t (64 keys = items 2t, 2t+1): V^T(t) and K(t+2) are staged at the start of its step; P.V runs one tile behind
// QK^T, so V^T(t) is first read in step t+1 and the round-1 ring (K 3 slots, V^T 2 slots, 80 KiB) gives K two steps and
// V one step of lead; `s_waitcnt vmcnt(4)` + s_barrier end a step.
// Lazy integer running max without a first-tile case (m~ starts at 0, moves up or down by integers in the exact
// path), whole-row sums in both half-lanes, dtype-dependent lower threshold: as in bsattn2.hip.
//
// The steady state of an image query block runs in a main loop unrolled over the six-step period of the two rings
// (LP_STEP_C): ring slots are compile-time constants there and fold into the ds_read offset fields; the K fragments of
// a tile's second half are requested under the first half's P.V MFMAs (PRE); the four LDS-DMA pieces of a stage go out
// in MFMA slots 1 / 5 / 9 / 13 of a block; the kept-list window is checked once per six steps.  A wave's instruction
// stream is in-order, so each of these took instructions out of the gaps between its MFMAs: MFMA-pipe busy 64 % -> 79 %
// (profiles/r02_pmc_bsattn_lp*.json).  The first step, the < 6 remainder steps, the tail blocks that need text_amp or
// the kv-length mask, and the text rows use the generic forms (LP_STEP, lp_slow_tile).
//
// Launch modes (instantiations of one kernel template, see bsattn_lp_kernel): the static mapping of query blocks to
// workgroups (what a capturing stream gets), the BALANCED launch (default: workgroups draw their query block from per-XCD
// queues -- the XCDs of a chip run 3-8 % apart; bit-identical to the static mapping) and the dense cross-attention of the Wan
// blocks.  Two rules that the measurements behind them produced, both checked by tests/test_isa_cpu.py:
//   * the kernel sits at 256 VGPRs; anything compiled into the static instantiation moves spill reloads into the unrolled
//     main loop -- new modes are new instantiations;
//   * nothing the compiler can take for a store (an atomic, s_sleep, s_memrealtime) in front of the main loop: the uniform
//     loads behind it stop being scalar loads, a DMA offset goes to scratch, and every reload drains the DMA queue.
// The launch-order experiments of rounds 3-4 (cohort start barrier, rotated list walk on a clock cursor and its replay /
// position modes) are measured, recorded in profiles/r04_attn_*.json and DESIGN.md, and gone from the source (git 882bacf).
#include <cstdlib>

#include "lp_balance.h"
#include "lp_core.h"

namespace jenga {
namespace {

struct LpParams {
    const uint16_t* q;
    const uint16_t* k;
    const uint16_t* vt;
    uint16_t* o;
    const int32_t* seqlens;
    const int32_t* idx;
    const int32_t* cnt;
    const int32_t* order;   // optional launch-order hint: position -> query block, per (b, h) (jenga_order_by_count)
    long long q_sb, q_ss, q_sh, k_sb, k_ss, k_sh, o_sb, o_ss, o_sh;
    int B, H, n_blocks, nq_img;
    int text_block_start;
    float qk_scale;
    float text_amp;
    int n_text_q;    // query blocks that run in TEXT mode (all kv blocks, no list, no mask) and the first of them
    int q_text0;
    int text_kv_len; // > 0 (jenga_cross_attn_fwd): TEXT-mode rows see keys < text_kv_len only (the last tile may be ragged)
    int n_text_wg_pad;
    int img_per_head;
    int xcd_chunk;
    int bal_set;     // JENGA_ATTN_BALANCE: which of the LP_BAL_SETS ticket-counter sets this launch draws from
};

// XKV: the cross-attention instantiation (TEXT rows against a kv sequence whose last tile may be ragged); a template
// parameter so that the product kernel's code (and its register allocation) is exactly what it is without that path
template <typename T, bool TEXT, bool XKV = false>
__device__ __forceinline__ void attn_block_lp(const LpParams& P, unsigned char* smem, int b, int h, int m) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 31, hi = lane >> 5;
    const int seqlen = P.seqlens ? __builtin_amdgcn_readfirstlane(P.seqlens[b]) : P.n_blocks * 128;

    const int32_t* list = nullptr;
    int nkept;
    if (TEXT) {
        nkept = P.n_blocks;
    } else {
        const long long row = ((long long)b * P.H + h) * P.nq_img + m;
        list = P.idx + row * P.n_blocks;
        nkept = __builtin_amdgcn_readfirstlane(P.cnt[row]);
    }

    LpState st;
    const long long qrow = (long long)m * 128 + wave_u * 32 + lq;
    {
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
            st.qf[ds] = raw;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) st.o[i][r] = 0.f;
    st.l = 0.f;
    st.neg_m = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) st.cinit[r] = 0.f;
    uint16_t* const op = P.o + b * P.o_sb + qrow * P.o_ss + h * P.o_sh + hi * 4;

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

    // kept list, 64 entries at a time in one VGPR (bsattn.hip)
    int lchunk = 0, lbase = -64;
    auto blk_at = [&](int i) -> int {
        if (i >= nkept) i = nkept - 1;   // the last steps stage one (unused) tile more: same piece count every step
        if (TEXT) return i;
        if (i < lbase || i >= lbase + 64) {
            lbase = i & ~63;
            lchunk = (lbase + lane < nkept) ? list[lbase + lane] : 0;
            // wait HERE for the (rare) reload: at the join in front of v_readlane hipcc's vmcnt(0) runs every step and
            // drains the whole LDS-DMA prefetch (the hardware counter includes the asm loads)
            __builtin_amdgcn_s_waitcnt(0x0F70);
        }
        return __builtin_amdgcn_readlane(lchunk, i - lbase);
    };
    // tile t = half (t & 1) of kept block t >> 1
    auto issue_k_at = [&](int t, int slot) {
        const int tc = t < 2 * nkept ? t : 2 * nkept - 1;
        const int blk = blk_at(tc >> 1);
        // byte offset = row * (k_ss * 2): one 32 x 32 -> 64 bit scalar multiply (launcher: k_ss * 2 < 2^32)
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
    // unrolled main loop: the 64-entry window is checked ONCE per six steps (lp_window), the per-stage lookups are a
    // bare v_readlane -- the per-lookup check costs ~10 scalar instructions in front of every block
    auto lp_window = [&](int t) {
        if (TEXT) return;
        const int first = t >> 1, last = (t + 7) >> 1;
        if (first < lbase || last >= lbase + 64) {
            lbase = first;
            lchunk = (lbase + lane < nkept) ? list[lbase + lane] : 0;
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
    auto issue_v = [&](int t) { issue_v_at(t, t & 1); };
    auto kslot = [&](int t) { return smem + (t % 3) * LP_TILE; };   // + k_addr (LP_K_RING inside)
    auto vslot = [&](int t) { return smem + (t & 1) * LP_TILE; };   // + v_addr (LP_V_RING inside)

    // blocks at the tail of the ascending list that need the text_amp / kv-length path
    int n_fast = nkept;
    if (!TEXT) {
        while (n_fast > 0) {
            const int bl = blk_at(n_fast - 1);
            if (bl >= P.text_block_start || (bl + 1) * 128 > seqlen) --n_fast; else break;
        }
    }
    // tiles lying entirely behind the kv length contribute exp2(-inf) = 0: not staged at all (the list is ascending,
    // so they are its last tiles -- with 64 valid text tokens: the second half of text block 0 and all of block 1)
    int t_all = 2 * nkept;
    if (!TEXT) {
        while (t_all > 0 && blk_at((t_all - 1) >> 1) * 128 + ((t_all - 1) & 1) * 64 >= seqlen) --t_all;
    }
    int t_fast = 2 * n_fast < t_all ? 2 * n_fast : t_all;
    if (XKV && P.text_kv_len > 0) {     // cross-attention: whole tiles in the pipeline, the ragged one in the slow form
        t_all = (P.text_kv_len + 63) >> 6;
        t_fast = P.text_kv_len >> 6;
    }

    f32x16 sA, sB;
    uint4 pfA[2], pfB[2];
    uint4 frk[8];   // K fragments of the item in flight (lp_bb, PRE)
#define LP_PRE0 1
#define LP_PRE1 2
#pragma unroll
    for (int r = 0; r < 16; ++r) sA[r] = sB[r] = 0.f;
    pfA[0] = pfA[1] = pfB[0] = pfB[1] = make_uint4(0u, 0u, 0u, 0u);

    // prologue: K(0), K(1)
    if (nkept > 0) {
        issue_k(0);
        issue_k(1);
    }
    LP_WAIT_ALL();
    __syncthreads();

    // step t: stage V^T(t) and K(t+2); QK^T on K(t); P.V on V^T(t-1)
#define LP_STEP(T_, PV_, SM0_)                                                                                        \
    do {                                                                                                              \
        issue_v(T_);                                                                                                  \
        lp_bb<T, TEXT, 0, PV_, true, SM0_>(st, kslot(T_), vslot((T_) - 1), sA, sB, pfA, pfB, k_addr, v_addr,          \
                                           P.qk_scale, frk);                                                          \
        issue_k((T_) + 2);                                                                                            \
        lp_bb<T, TEXT, 1, PV_, true, true>(st, kslot(T_), vslot((T_) - 1), sB, sA, pfB, pfA, k_addr, v_addr,          \
                                           P.qk_scale, frk);                                                          \
        LP_WAIT_KEEP4();                                                                                              \
        __syncthreads();                                                                                              \
    } while (0)
    // step t0 + J of the unrolled loop, t0 = 1 (mod 6): every ring slot is a compile-time constant
#define LP_STEP_C(T0_, J_)                                                                                            \
    do {                                                                                                              \
        {                                                                                                             \
            const LpDma dv_ = desc_v_at((T0_) + (J_), (1 + (J_)) & 1);                                                \
            lp_bb<T, TEXT, 0, true, true, true, ((1 + (J_)) % 3) * LP_TILE, ((J_) & 1) * LP_TILE, LP_PRE0>(           \
                st, smem, smem, sA, sB, pfA, pfB, k_addr, v_addr, P.qk_scale, frk, &dv_);                             \
        }                                                                                                             \
        {                                                                                                             \
            const LpDma dk_ = desc_k_at((T0_) + (J_) + 2, (J_) % 3);                                                  \
            lp_bb<T, TEXT, 1, true, true, true, ((1 + (J_)) % 3) * LP_TILE, ((J_) & 1) * LP_TILE, LP_PRE1>(           \
                st, smem, smem, sB, sA, pfB, pfA, k_addr, v_addr, P.qk_scale, frk, &dk_);                             \
        }                                                                                                             \
        LP_WAIT_KEEP4();                                                                                              \
        __syncthreads();                                                                                              \
    } while (0)
    if (t_fast > 0) {
        LP_STEP(0, false, false);
        int t = 1;
        if (!TEXT) {
            for (; t + 6 <= t_fast; t += 6) {
                lp_window(t);
                LP_STEP_C(t, 0); LP_STEP_C(t, 1); LP_STEP_C(t, 2); LP_STEP_C(t, 3); LP_STEP_C(t, 4); LP_STEP_C(t, 5);
            }
        }
        for (; t < t_fast; ++t) LP_STEP(t, true, true);
        // drain: softmax of the last item, P.V of the last tile
        lp_bb<T, TEXT, 0, true, false, true>(st, nullptr, vslot(t_fast - 1), sA, sB, pfA, pfB, k_addr, v_addr,
                                             P.qk_scale, frk);
        lp_bb<T, TEXT, 1, true, false, false>(st, nullptr, vslot(t_fast - 1), sB, sA, pfB, pfA, k_addr, v_addr,
                                              P.qk_scale, frk);
    }
#undef LP_STEP
#undef LP_STEP_C
#undef LP_PRE0
#undef LP_PRE1
    if (!TEXT || (XKV && P.text_kv_len > 0)) {
        for (int t = t_fast; t < t_all; ++t) {
            issue_v(t);
            issue_k(t + 2);
            LP_WAIT_ALL();   // this step reads V^T(t) itself
            __syncthreads();
            const int blk = blk_at(t >> 1);
            if (TEXT)
                lp_slow_tile<T, true>(st, kslot(t), vslot(t), blk * 128 + (t & 1) * 64, false, 0.f, P.text_kv_len, hi,
                                      k_addr, v_addr, P.qk_scale);
            else
                lp_slow_tile<T>(st, kslot(t), vslot(t), blk * 128 + (t & 1) * 64, blk >= P.text_block_start, P.text_amp,
                                seqlen, hi, k_addr, v_addr);
            __syncthreads();
        }
    }
    LP_WAIT_ALL();

    // ---- epilogue: o = acc / l, rows >= seqlen written as zeros (image rows only) ----
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
}

#define LP_THREADS 256
// VARIANT 0: the product kernel, static mapping.  2: dense cross-attention (jenga_cross_attn_fwd): TEXT-mode rows only,
// kv-length mask on the last tile.  4: query blocks drawn from per-XCD queues (JENGA_ATTN_BALANCE, lp_balance.h).  Separate
// instantiations on purpose (see the header).  (Variants 1, 3, 5, 6 were the launch-order experiments of round 4.)
template <typename T, int VARIANT>
__global__ void __launch_bounds__(LP_THREADS, 2) bsattn_lp_kernel(LpParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int n_text = P.n_text_q;
    const int id = blockIdx.x;
    if (id < P.n_text_wg_pad) {   // text query blocks first: the longest work items start earliest
        if (id >= P.B * P.H * n_text) return;
        const int m = P.q_text0 + id % n_text;
        const int bh = id / n_text;
        attn_block_lp<T, true, VARIANT == 2>(P, smem, bh / P.H, bh % P.H, m);
        return;
    }
    if (VARIANT == 2) return;
    int li = id - P.n_text_wg_pad;
    if (VARIANT == 4) {
        // the ticket becomes the launch position the static mapping would have given that block, and the code below runs
        // unchanged
        const int ticket = lp_draw_ticket(g_balance_ctr[P.bal_set], P.B * P.H, P.nq_img, P.xcd_chunk, li & 7,
                                          reinterpret_cast<int*>(smem));
        if (ticket < 0) return;
        const int y = ticket >> 28, t = ticket & 0x0fffffff;
        int nv = P.nq_img - y * P.xcd_chunk;
        nv = nv < P.xcd_chunk ? nv : P.xcd_chunk;
        li = __builtin_amdgcn_readfirstlane((t / nv) * P.img_per_head + (((t % nv) << 3) | y));
    }
    const int bh = li / P.img_per_head;
    const int r = li % P.img_per_head;
    int m;
    if (P.xcd_chunk) {
        m = (r & 7) * P.xcd_chunk + (r >> 3);
        if ((r >> 3) >= P.xcd_chunk || m >= P.nq_img) return;
    } else {
        m = r;
    }
    if (P.order) m = P.order[(long long)bh * P.nq_img + m];   // kept-count-aware order inside the XCD's range
    attn_block_lp<T, false>(P, smem, bh / P.H, bh % P.H, m);
}

template <typename T, int VARIANT>
static hipError_t lp_launch(const LpParams& P, long long grid, hipStream_t stream) {
    static bool smem_set[64] = {};
    lp_set_smem_once((const void*)bsattn_lp_kernel<T, VARIANT>, LP_LDS_BYTES, smem_set);
    hipLaunchKernelGGL((bsattn_lp_kernel<T, VARIANT>), dim3((unsigned)grid), dim3(LP_THREADS), LP_LDS_BYTES, stream, P);
    return hipGetLastError();
}

}  // namespace
}  // namespace jenga

using namespace jenga;

// same arguments as jenga_bsattn_fwd (bsattn.hip); reached through it with JENGA_ATTN_LP
int jenga_bsattn_lp_launch(void* stream, const void* q, const void* k, const void* vt, void* o, const int32_t* seqlens,
                           const int32_t* idx, const int32_t* cnt, const int32_t* order, int64_t B, int64_t H,
                           int64_t n_blocks, int64_t nq_img, int64_t q_sb, int64_t q_ss, int64_t q_sh, int64_t k_sb,
                           int64_t k_ss, int64_t k_sh, int64_t o_sb, int64_t o_ss, int64_t o_sh, float sm_scale,
                           float text_amp, int64_t text_block_start, int dtype, int flags) {
    LpParams P;
    P.q = (const uint16_t*)q;
    P.k = (const uint16_t*)k;
    P.vt = (const uint16_t*)vt;
    P.o = (uint16_t*)o;
    P.seqlens = seqlens;
    P.idx = idx;
    P.cnt = cnt;
    P.order = order;
    P.q_sb = q_sb; P.q_ss = q_ss; P.q_sh = q_sh;
    P.k_sb = k_sb; P.k_ss = k_ss; P.k_sh = k_sh;
    P.o_sb = o_sb; P.o_ss = o_ss; P.o_sh = o_sh;
    P.B = (int)B; P.H = (int)H; P.n_blocks = (int)n_blocks; P.nq_img = (int)nq_img;
    P.text_block_start = (int)text_block_start;
    P.qk_scale = (float)((double)sm_scale * 1.44269504);
    P.text_amp = text_amp;
    const long long n_text = n_blocks - nq_img;
    const long long n_text_wg = B * H * n_text;
    P.n_text_q = (int)n_text;
    P.q_text0 = (int)nq_img;
    P.text_kv_len = 0;
    P.bal_set = 0;
    P.n_text_wg_pad = (int)((n_text_wg + 7) / 8 * 8);
    if ((flags & JENGA_ATTN_XCD_REMAP) && nq_img >= 64) {
        P.xcd_chunk = (int)((nq_img + 7) / 8);
        P.img_per_head = P.xcd_chunk * 8;
    } else {
        P.xcd_chunk = 0;
        P.img_per_head = (int)nq_img;
    }
    if (k_ss < 0 || k_ss >= (1LL << 31)) {
        set_error("jenga_bsattn_fwd: k token stride %lld out of range", (long long)k_ss);
        return JENGA_EINVAL;
    }
    const long long grid = (long long)P.n_text_wg_pad + B * H * (long long)P.img_per_head;
    if (grid <= 0 || grid > 0x7fffffffLL) {
        set_error("jenga_bsattn_fwd: grid size %lld out of range", grid);
        return JENGA_EINVAL;
    }
    // the balanced launch keeps per-device state that is ordered by events: that cannot be recorded into a HIP graph, so a
    // capturing stream gets the plain static launch
    if (flags & JENGA_ATTN_BALANCE) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing((hipStream_t)stream, &cap) != hipSuccess) {
            (void)hipGetLastError();      // (the query itself failed: not this launch's error)
            cap = hipStreamCaptureStatusActive;
        }
        if (cap != hipStreamCaptureStatusNone) flags &= ~JENGA_ATTN_BALANCE;
    }
    // JENGA_ATTN_BALANCE: query blocks drawn from per-XCD queues, grid oversubscribed by 1/8 (JENGA_BALANCE_EXTRA_PCT, at
    // least 8 workgroups) so that a fast XCD has workgroups left to draw from a slow one's queue
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
        e = dtype == JENGA_BF16 ? lp_launch<BF16, 4>(P, grid_bal, (hipStream_t)stream)
                                : lp_launch<FP16, 4>(P, grid_bal, (hipStream_t)stream);
    else
        e = dtype == JENGA_BF16 ? lp_launch<BF16, 0>(P, grid, (hipStream_t)stream)
                                : lp_launch<FP16, 0>(P, grid, (hipStream_t)stream);
    if (bal_slot >= 0) lp_balance_release(bal_slot, (hipStream_t)stream);
    if (e != hipSuccess) {
        set_error("jenga_bsattn_fwd (lp): %s", hipGetErrorString(e));
        return JENGA_ELAUNCH;
    }
    return JENGA_OK;
}

// Dense cross-attention on the same kernel: every query block runs in TEXT mode (all kv blocks, fp32 scores x
// sm_scale * log2 e, no kv-length mask, no list) against a kv sequence of its OWN length.  Replaces the flash_attention
// call of WanT2VCrossAttention (wan/modules/model_mul.py:183-205: 512 context tokens, k_lens = None).
extern "C" int jenga_cross_attn_fwd(void* stream, const void* q, const void* k, const void* vt, void* o, int64_t B,
                                    int64_t H, int64_t nq_blocks, int64_t nkv_blocks, int64_t kv_len, int64_t q_sb, int64_t q_ss,
                                    int64_t q_sh, int64_t k_sb, int64_t k_ss, int64_t k_sh, int64_t o_sb, int64_t o_ss,
                                    int64_t o_sh, float sm_scale, int dtype) {
    auto bad8 = [](int64_t a, int64_t b_, int64_t c) { return (a & 7) || (b_ & 7) || (c & 7); };
    if (!q || !k || !vt || !o || B <= 0 || H <= 0 || nq_blocks <= 0 || nkv_blocks <= 0 || kv_len <= 0 ||
        kv_len > nkv_blocks * 128 || kv_len <= (nkv_blocks - 1) * 128 || bad8(q_sb, q_ss, q_sh) ||
        bad8(k_sb, k_ss, k_sh) || bad8(o_sb, o_ss, o_sh) || ((uintptr_t)q & 15) || ((uintptr_t)k & 15) ||
        ((uintptr_t)vt & 15) || ((uintptr_t)o & 7) || k_ss < 0 || k_ss >= (1LL << 31)) {
        set_error("jenga_cross_attn_fwd: bad arguments (non-null 16-B aligned pointers, strides multiples of 8 elements, "
                  "at least one query block, kv_len inside the last of the nkv_blocks kv blocks)");
        return JENGA_EINVAL;
    }
    if (dtype != JENGA_BF16 && dtype != JENGA_FP16) {
        set_error("jenga_cross_attn_fwd: dtype must be bf16 or fp16");
        return JENGA_EUNSUPPORTED;
    }
    LpParams P;
    P.q = (const uint16_t*)q; P.k = (const uint16_t*)k; P.vt = (const uint16_t*)vt; P.o = (uint16_t*)o;
    P.seqlens = nullptr; P.idx = nullptr; P.cnt = nullptr; P.order = nullptr;
    P.q_sb = q_sb; P.q_ss = q_ss; P.q_sh = q_sh;
    P.k_sb = k_sb; P.k_ss = k_ss; P.k_sh = k_sh;
    P.o_sb = o_sb; P.o_ss = o_ss; P.o_sh = o_sh;
    P.B = (int)B; P.H = (int)H; P.n_blocks = (int)nkv_blocks; P.nq_img = 0;
    P.text_block_start = (int)nkv_blocks;
    P.qk_scale = (float)((double)sm_scale * 1.44269504);
    P.text_amp = 0.f;
    P.n_text_q = (int)nq_blocks;
    P.q_text0 = 0;
    P.text_kv_len = (kv_len == nkv_blocks * 128) ? 0 : (int)kv_len;     // 0: every tile whole, no mask needed
    const long long wg = B * H * nq_blocks;
    P.n_text_wg_pad = (int)((wg + 7) / 8 * 8);
    P.xcd_chunk = 0;
    P.img_per_head = 0;
    P.bal_set = 0;
    if (wg > 0x7ffffff0LL) {
        set_error("jenga_cross_attn_fwd: grid size %lld out of range", wg);
        return JENGA_EINVAL;
    }
    const hipError_t e = dtype == JENGA_BF16 ? lp_launch<BF16, 2>(P, P.n_text_wg_pad, (hipStream_t)stream)
                                             : lp_launch<FP16, 2>(P, P.n_text_wg_pad, (hipStream_t)stream);
    if (e != hipSuccess) {
        set_error("jenga_cross_attn_fwd: %s", hipGetErrorString(e));
        return JENGA_ELAUNCH;
    }
    return JENGA_OK;
}
