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
// Launch modes (round 4; instantiations of one kernel template, see bsattn_lp_kernel): the static mapping of query blocks
// to workgroups (round 3's launch; what a capturing stream gets), the BALANCED launch (default: workgroups draw their query
// block from per-XCD queues -- the XCDs of a chip run 3-8 % apart), the rotated list walk on a clock cursor (opt-in, not
// bit-reproducible) on either mapping, and the dense cross-attention of the Wan blocks.  Two rules that the measurements
// behind them produced, both checked by tests/test_isa_cpu.py:
//   * the kernel sits at 256 VGPRs; anything compiled into the static instantiation moves spill reloads into the unrolled
//     main loop -- new modes are new instantiations;
//   * nothing the compiler can take for a store (an atomic, s_sleep, s_memrealtime) in front of the main loop: the uniform
//     loads behind it stop being scalar loads, a DMA offset goes to scratch, and every reload drains the DMA queue.
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

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

// LP_EXP: the measured-and-rejected launch modes (cohort start barrier; position / replay modes of the rotated walk;
// per-workgroup tick dump) are compiled into libjenga_amd_exp.so only (python -m jenga_amd.build --experiments)
#ifdef JENGA_EXPERIMENTS
#define LP_EXP 1
#else
#define LP_EXP 0
#endif
#if LP_EXP
#define LP_ROT_REPLAY (-1000000000)   // rot_period value selecting the replay branch
__device__ unsigned short* g_rot_table;   // JENGA_ROTATE_REPLAY: one start phase (16-bit fraction of a turn) per image workgroup
__device__ unsigned* g_rot_times;          // record mode: [start, end] wall-clock ticks (low 32 bits) per launch position
__device__ int g_rot_table_mode;          // 0 off, 1 record (clock mode writes the phase it used), 2 replay (read instead of the clock)
__device__ float g_rot_spread = 0.42f;  // JENGA_ATTN_ROTATE position mode: growth of the start-time spread per sqrt(generation)
#endif

// XKV: the cross-attention instantiation (TEXT rows against a kv sequence whose last tile may be ragged); a template
// parameter so that the product kernel's code (and its register allocation) is exactly what it is without that path
// ROT (round 4, JENGA_ATTN_ROTATE; 2 = clock mode, 1 = clock / replay / position modes in the experiments library -- a separate
// instantiation: the balanced launch's main loop keeps its DMA offsets in registers only without the other modes' code): the fast part of the ascending list is walked from a ROTATED start --
// logical entry j < n_rot is physical entry (j + rot) mod n_rot, rot = phase of a chip-wide wall-clock cursor x n_rot -- so
// that workgroups started at different times are at the same kv blocks at the same time WITHOUT waiting for each other.
// The summation order of the online softmax then depends on the start time: results are equal within fp32 rounding of
// the running sums, not bit-identical from run to run.  rot_period: the cursor's period in wall-clock ticks.
template <typename T, bool TEXT, bool XKV = false, int ROT = 0>
__device__ __forceinline__ void attn_block_lp(const LpParams& P, unsigned char* smem, int b, int h, int m,
                                              int rot_period = 0, int rot_seq = 0) {
    (void)rot_seq;      // (the experiments library's position / replay modes index by it)
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
    int rot = 0, n_rot = 0;
    auto phys = [&](int j) -> int {
        if (!ROT || j >= n_rot) return j;
        const int p_ = j + rot;
        return p_ >= n_rot ? p_ - n_rot : p_;
    };
    auto blk_at = [&](int i) -> int {
        if (i >= nkept) i = nkept - 1;   // the last steps stage one (unused) tile more: same piece count every step
        if (TEXT) return i;
        if (i < lbase || i >= lbase + 64) {
            lbase = i & ~63;
            // (ROT: the window lives in LOGICAL index space, every lane fetches its own physical entry -- the rotation's
            // wrap point needs no special case anywhere else)
            lchunk = (lbase + lane < nkept) ? list[phys(lbase + lane)] : 0;
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
            lchunk = (lbase + lane < nkept) ? list[phys(lbase + lane)] : 0;
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
    if (ROT && n_fast > 1 && rot_period > 0) {
        // clock mode: one clock reading for the whole workgroup (its four waves share the tile order)
        if (tid == 0) {
            const unsigned long long c = (unsigned long long)wall_clock64() % (unsigned long long)rot_period;
            *reinterpret_cast<int*>(smem) = (int)((c * (unsigned long long)n_fast) / (unsigned long long)rot_period);
#if LP_EXP
            if (ROT == 1 && g_rot_table_mode == 1)      // record: the phase this workgroup started at (rot_seq = its launch position)
                g_rot_table[rot_seq] = (unsigned short)((c << 16) / (unsigned long long)rot_period);
#endif
        }
        __syncthreads();
        rot = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int*>(smem));
        n_rot = n_fast;
        lbase = -64;          // (the window holds unrotated entries from the tail scan)
        __syncthreads();
#if LP_EXP
    } else if (ROT == 1 && n_fast > 1 && rot_period == LP_ROT_REPLAY) {
        // replay (deterministic): the start phase a clock-mode launch of this shape recorded for this launch position
        rot = (int)(((unsigned)g_rot_table[rot_seq] * (unsigned)n_fast) >> 16);
        n_rot = n_fast;
        lbase = -64;
    } else if (ROT == 1 && n_fast > 1 && rot_period < 0) {
        // position mode (deterministic): in the steady state an XCD starts S = -rot_period workgroups per mean workgroup
        // lifetime (work conservation: its S slots are always full), so the rot_seq-th workgroup of the XCD's queue starts
        // at cursor phase frac(rot_seq / S) -- no clock, no period to know
        const int S = -rot_period;
        // ... once the starts are staggered.  A launch begins with all S slots starting TOGETHER (true phase 0 for all of
        // them); the stagger then grows generation by generation as lifetimes vary (a random walk per slot: the width of
        // the start-time spread after g generations is ~ spread * sqrt(g) of a lifetime, capped at one lifetime).  The
        // k-th workgroup to start in generation g therefore sits at phase (k / S - 1/2) * min(1, spread * sqrt(g)).
        const int g = rot_seq / S, k = rot_seq % S;
        float w = g_rot_spread * __builtin_sqrtf((float)g);
        w = w > 1.f ? 1.f : w;
        float ph = ((float)k / (float)S - 0.5f) * w;
        ph -= __builtin_floorf(ph);
        rot = (int)(ph * (float)n_fast);
        rot = rot >= n_fast ? n_fast - 1 : rot;
        n_rot = n_fast;
        lbase = -64;
#endif
    }
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

#if LP_EXP
// JENGA_ATTN_COHORT (round-4 experiment): the workgroups an XCD runs at a time start TOGETHER -- one arrival counter per
// XCD and per generation of `size` consecutive launch positions, bounded spin -- so that they walk their ascending lists
// in step and meet in the XCD's L2.  The configuration lives in a device global written on the launch stream, NOT in
// LpParams: the kernel sits at 256 VGPRs and a longer kernarg block moves spill reloads into the unrolled main loop.
struct CohortCfg {
    int* ctr;
    int size, stride, timeout;   // timeout in wall-clock ticks (100 MHz)
    int quorum;                  // arrivals a member waits for (<= size): the stragglers of a generation start late
};
__device__ CohortCfg g_cohort_cfg;
#endif
__device__ int g_rot_period_ticks;      // JENGA_ATTN_ROTATE: the cursor's period (wall-clock ticks, 100 MHz)
#define LP_BAL_SETS 64
__device__ int g_balance_ctr[LP_BAL_SETS][8];   // JENGA_ATTN_BALANCE: tickets drawn from each XCD's queue, one set per launch in
                                               // flight (the launcher hands the sets out in turn and zeroes one on the stream)
__device__ int g_rot_T_est = 87500;     // lifetime of the image workgroup that finished last (ticks): the NEXT launch's period
                                        // in auto mode (copied device-to-device on the launch stream; 875 us to begin with)

// instrumentation (experiments library, JENGA_LP_TIMES_DUMP): [start, end] of every image workgroup of the rotated variants
__device__ __forceinline__ void lp_record_times(int li, long long t_start, long long t_end) {
#if LP_EXP
    unsigned* tm = g_rot_times;
    if (tm) {
        tm[2 * li] = (unsigned)t_start;
        tm[2 * li + 1] = (unsigned)t_end;
    }
#else
    (void)li; (void)t_start; (void)t_end;
#endif
}

// JENGA_ATTN_BALANCE: thread 0 draws (queue y, ticket t) -- own queue first, then the fullest other one, at most 8 attempts --
// and the workgroup gets it through LDS as (y << 28 | t), or -1 when every queue is empty.
// The counter accesses are inline assembly WITHOUT a memory clobber, on purpose: an atomic the compiler can see counts as a
// possible write to everything the kernel loads afterwards, its uniform loads (launch order, kept count, sequence length,
// the list window) stop being scalar loads and come back through VGPRs, and with 256 VGPRs in use that costs the main loop
// one or two of its DMA offsets -- reloaded from scratch three times per 12 steps, each reload draining the DMA queue
// (measured: -2.8 % before any balancing gain).  Nothing else in the kernel reads or writes the counters.
typedef int lp_int4 __attribute__((ext_vector_type(4)));
// (not `volatile`, no memory clobber: to the compiler these are pure functions of their operands -- `seq` differs between
// any two calls of a workgroup so that they are never merged)
__device__ __forceinline__ int lp_ticket_add(int* p, int seq) {   // p: wave-uniform
    int old;
    const unsigned zero = 0;
    const int one = 1;
    asm("global_atomic_add %0, %1, %2, %3 sc0\n\ts_waitcnt vmcnt(0) ; draw %4" : "=v"(old) : "v"(zero), "v"(one), "s"(p), "s"(seq));
    return old;
}
__device__ __forceinline__ int lp_draw_ticket(int* ctr, int BH, int nq_img, int xcd_chunk, int x, int* lds) {
    if (threadIdx.x == 0) {
        auto qlen = [&](int z) {
            int nv = nq_img - z * xcd_chunk;
            nv = nv < xcd_chunk ? nv : xcd_chunk;
            return nv > 0 ? BH * nv : 0;
        };
        int y = x, t = qlen(x);
        if (t > 0) t = lp_ticket_add(ctr + x, -1);
        if (t >= qlen(x)) {
            y = -1;
#pragma nounroll
            for (int attempt = 0; attempt < 8 && y < 0; ++attempt) {
                lp_int4 c0, c1;
                const unsigned zero = 0;
                asm("global_load_dwordx4 %0, %2, %3 sc1\n\tglobal_load_dwordx4 %1, %2, %3 offset:16 sc1\n\t"
                    "s_waitcnt vmcnt(0) ; scan %4"
                    : "=&v"(c0), "=&v"(c1)
                    : "v"(zero), "s"(ctr), "s"(attempt));
                const int drawn[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
                int best = -1, left = 0;
#pragma unroll
                for (int z = 0; z < 8; ++z) {
                    const int l = qlen(z) - drawn[z];
                    if (l > left) { left = l; best = z; }
                }
                if (best < 0) break;
                best = __builtin_amdgcn_readfirstlane(best);     // (one lane is active: its value, in an SGPR for the asm)
                const int tt = lp_ticket_add(ctr + best, attempt);
                if (tt < qlen(best)) { y = best; t = tt; }
            }
        }
        lds[0] = y < 0 ? -1 : (y << 28 | t);
    }
    __syncthreads();
    const int ticket = __builtin_amdgcn_readfirstlane(lds[0]);
    __syncthreads();
    return ticket;
}

#define LP_THREADS 256
// VARIANT 0: the product kernel.  1: + the cohort start barrier.  2: dense cross-attention (jenga_cross_attn_fwd):
// TEXT-mode rows only, kv-length mask on the last tile.  3: rotated list walk (JENGA_ATTN_ROTATE).  4: query blocks drawn
// from per-XCD queues (JENGA_ATTN_BALANCE).  5: 4 + 3.  6 (experiments library): 4 + 1.  Separate instantiations on purpose (see above).  (Two more were measured and removed: rotation + pacing -- a workgroup ahead of the
// cursor sleeps -- lost 10 %; rotation + whole heads per XCD got the L2 hit rate to 47 % and 35.5 KB per kept pair but ran
// 2 % behind plain rotation: eight heads in flight overflow the Infinity Cache.  profiles/r04_attn_rotate_ab.json.)
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
    int bal_seq = 0, bal_total = 0;
    if (VARIANT == 4 || VARIANT == 5 || VARIANT == 6) {
        // cross-XCD balancing: the hardware deals workgroup ids to the 8 XCDs round robin, so with one workgroup per query
        // block every XCD gets the same number of blocks whatever its speed -- and the XCDs of one chip differ by several
        // per cent.  Here a workgroup DRAWS its query block: a ticket from the queue of the XCD it runs on (the same
        // contiguous range, the same order as the static mapping), and once that queue is empty from the queue with the
        // most blocks left.  The grid is oversubscribed (the launcher adds 1/8) so that a fast XCD has workgroups left
        // to draw with; a workgroup that finds every queue empty exits.  Which workgroup computes a block does not enter
        // the result: bit-identical to the static mapping.  The ticket becomes the launch position the static mapping
        // would have given that block, and the code below runs unchanged.
        const int ticket = lp_draw_ticket(g_balance_ctr[P.bal_set], P.B * P.H, P.nq_img, P.xcd_chunk, li & 7,
                                          reinterpret_cast<int*>(smem));
        if (ticket < 0) return;
        const int y = ticket >> 28, t = ticket & 0x0fffffff;
        int nv = P.nq_img - y * P.xcd_chunk;
        nv = nv < P.xcd_chunk ? nv : P.xcd_chunk;
        li = __builtin_amdgcn_readfirstlane((t / nv) * P.img_per_head + (((t % nv) << 3) | y));
        bal_seq = t;
        bal_total = __builtin_amdgcn_readfirstlane(P.B * P.H * nv);
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
#if LP_EXP
    if (VARIANT == 1 || VARIANT == 6) {      // (6: on drawn blocks -- the remapped launch position IS the queue position)
        if (threadIdx.x == 0) {
            const CohortCfg C = g_cohort_cfg;
            const int x = r & 7, pos = r >> 3;
            // a workgroup that drew from another XCD's queue shares no L2 with that cohort: it reports in and does not wait
            const bool guest = VARIANT == 6 && (((int)blockIdx.x - P.n_text_wg_pad) & 7) != x;
            int nv = P.nq_img - x * P.xcd_chunk;                 // valid launch positions of this XCD per (b, h)
            nv = nv < P.xcd_chunk ? nv : P.xcd_chunk;
            const int seq = bh * nv + pos, total = P.B * P.H * nv;
            const int gen = seq / C.size;
            int members = total - gen * C.size;
            members = members < C.size ? members : C.size;
            members = members < C.quorum ? members : C.quorum;
            int* c = C.ctr + x * C.stride + gen;
            {   // (one lane is active: its pointer, in SGPRs for the asm operands)
                const unsigned long long cv = (unsigned long long)c;
                const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)cv);
                const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(cv >> 32));
                c = reinterpret_cast<int*>(((unsigned long long)hi << 32) | lo);
            }
            // (opaque accesses like the ticket code's: with the compiler's own atomics this variant's main loop reloaded
            // DMA offsets from scratch -- part of the loss recorded for it in profiles/r04_attn_cohort_ab.json)
            int arrived = lp_ticket_add(c, -2) + 1;
            auto now = [](int seq) {      // (s_memrealtime as a pure function of `seq`: the builtin counts as a memory access)
                unsigned long long t;
                asm("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0) ; clock %1" : "=s"(t) : "s"(seq));
                return (long long)t;
            };
            const long long t0 = now(-1);
            int spin = 0;
            while (!guest && arrived < members && now(spin) - t0 < C.timeout) {
                const unsigned zero = 0;
                asm("s_sleep 16\n\tglobal_load_dword %0, %1, %2 sc1\n\ts_waitcnt vmcnt(0) ; poll %3" : "=v"(arrived) : "v"(zero), "s"(c), "s"(spin));
                ++spin;
            }
        }
        __syncthreads();
    }
#endif
    if (P.order) m = P.order[(long long)bh * P.nq_img + m];   // kept-count-aware order inside the XCD's range
    if (VARIANT == 3 || VARIANT == 5) {
        int seq = li, seq_total = P.B * P.H * P.img_per_head;
        if (VARIANT == 5) {
            seq = bal_seq;
            seq_total = bal_total;
        } else if (P.xcd_chunk) {      // this workgroup's number in its XCD's queue (launch positions, before the count order)
            int nv = P.nq_img - (r & 7) * P.xcd_chunk;
            nv = nv < P.xcd_chunk ? nv : P.xcd_chunk;
            seq = bh * nv + (r >> 3);
            seq_total = P.B * P.H * nv;
        }
        const bool mid_queue = seq * 4 >= seq_total && seq * 4 < 3 * seq_total;
        int period = g_rot_period_ticks;
        if (period > 0) period = period < 5000 ? 5000 : (period > 1000000 ? 1000000 : period);   // 50 us .. 10 ms
#if LP_EXP
        if (VARIANT == 3) {
            const int table_mode = g_rot_table_mode;
            if (table_mode == 2) period = LP_ROT_REPLAY;
            if (table_mode) seq = li;        // the table is indexed by launch position
        }
#endif
        const long long t_start = (long long)wall_clock64();
        attn_block_lp<T, false, false, (LP_EXP && VARIANT == 3) ? 1 : 2>(P, smem, bh / P.H, bh % P.H, m, period, seq);
        // the next launch's period (auto mode): the lifetime of a workgroup from the MIDDLE of its XCD's queue -- the last
        // ones run on a draining chip and are faster than the steady state the cursor has to match
        if (threadIdx.x == 0) {
            const long long t_end = (long long)wall_clock64();
            if (mid_queue) g_rot_T_est = (int)(t_end - t_start);
            lp_record_times((int)blockIdx.x - P.n_text_wg_pad, t_start, t_end);
        }
    } else
        attn_block_lp<T, false>(P, smem, bh / P.H, bh % P.H, m);
}

template <typename T, int VARIANT>
static hipError_t lp_launch(const LpParams& P, long long grid, hipStream_t stream) {
    const size_t smem = LP_LDS_BYTES;
    (void)hipFuncSetAttribute((const void*)bsattn_lp_kernel<T, VARIANT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)smem);
    hipLaunchKernelGGL((bsattn_lp_kernel<T, VARIANT>), dim3((unsigned)grid), dim3(LP_THREADS), smem, stream, P);
    return hipGetLastError();
}

// ---- host side of the experiment variants (state per device; launches carrying these flags must not overlap on one device)

#if LP_EXP
// JENGA_LP_TIMES_DUMP=<file>: the rotate / balance variants write [start, end] wall-clock ticks (100 MHz, low 32 bits) of
// every image workgroup, indexed by launch position; the file holds the launch BEFORE the current one (written in front of
// the next launch with the variable set, after a stream synchronisation) as raw uint32 pairs.
void lp_times_hook(long long grid, hipStream_t stream) {
    static unsigned* times[64] = {nullptr};
    static long long times_n[64] = {0}, used_n[64] = {0};
    static bool armed[64] = {false};     // the device global holds a buffer (it starts out null)
    const char* file = getenv("JENGA_LP_TIMES_DUMP");
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return;
    if (!file && !armed[dev]) return;
    if (file && used_n[dev] > 0 && hipStreamSynchronize(stream) == hipSuccess) {
        std::vector<unsigned> h((size_t)used_n[dev] * 2, 0);
        if (hipMemcpy(h.data(), times[dev], h.size() * sizeof(unsigned), hipMemcpyDeviceToHost) == hipSuccess)
            if (FILE* f = fopen(file, "wb")) {
                fwrite(h.data(), sizeof(unsigned), h.size(), f);
                fclose(f);
            }
    }
    used_n[dev] = 0;
    unsigned* ptr = nullptr;
    if (file) {
        if (times_n[dev] < grid) {
            if (times[dev]) (void)hipFree(times[dev]);
            times[dev] = nullptr;
            times_n[dev] = hipMalloc((void**)&times[dev], (size_t)grid * 2 * sizeof(unsigned)) == hipSuccess ? grid : 0;
        }
        if (times_n[dev] >= grid && hipMemsetAsync(times[dev], 0, (size_t)grid * 2 * sizeof(unsigned), stream) == hipSuccess) {
            ptr = times[dev];
            used_n[dev] = grid;
        }
    }
    (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_rot_times), &ptr, sizeof(ptr), 0, hipMemcpyHostToDevice, stream);
    armed[dev] = ptr != nullptr;
}
#endif

// JENGA_ATTN_BALANCE: the ticket counters of a launch.  LP_BAL_SETS sets per device, handed out in turn under a mutex (ranks
// simulated by threads launch concurrently on one device); a set is zeroed on the launch stream in front of the kernel, and
// an event recorded behind the kernel makes the NEXT user of the set -- LP_BAL_SETS launches later, possibly on another
// stream -- wait for it, so two launches in flight never share counters.  (A capturing stream never gets here: the launcher
// drops the flag, an event recorded inside a capture cannot order a set against launches outside.)
struct LpBalanceSlots {
    std::mutex mu;
    int* base = nullptr;
    hipEvent_t done[LP_BAL_SETS] = {};
    bool used[LP_BAL_SETS] = {};
    bool busy[LP_BAL_SETS] = {};
    unsigned next = 0;
};
LpBalanceSlots g_bal[64];

int lp_balance_acquire(hipStream_t stream) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
    LpBalanceSlots& S = g_bal[dev];
    std::lock_guard<std::mutex> lock(S.mu);
    if (!S.base && hipGetSymbolAddress((void**)&S.base, HIP_SYMBOL(g_balance_ctr)) != hipSuccess) {
        S.base = nullptr;
        return -1;
    }
    int k = -1;
    for (int tries = 0; tries < LP_BAL_SETS && k < 0; ++tries) {     // (a set between acquire and release belongs to another thread)
        const int c = (int)(S.next++ % LP_BAL_SETS);
        if (!S.busy[c]) k = c;
    }
    if (k < 0) return -1;
    if (!S.done[k] && hipEventCreateWithFlags(&S.done[k], hipEventDisableTiming) != hipSuccess) {
        S.done[k] = nullptr;
        return -1;
    }
    if (S.used[k] && hipStreamWaitEvent(stream, S.done[k], 0) != hipSuccess) return -1;
    if (hipMemsetAsync(S.base + 8 * k, 0, 8 * sizeof(int), stream) != hipSuccess) return -1;
    S.busy[k] = true;
    return k;
}

void lp_balance_release(int k, hipStream_t stream) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return;
    LpBalanceSlots& S = g_bal[dev];
    std::lock_guard<std::mutex> lock(S.mu);
    S.used[k] = hipEventRecord(S.done[k], stream) == hipSuccess;
    if (!S.used[k]) (void)hipStreamSynchronize(stream);   // (no event: the set must be idle before anybody reuses it)
    S.busy[k] = false;
}

#if LP_EXP
// JENGA_ROTATE_REPLAY=record|replay: a clock-mode launch writes every workgroup's start phase to a per-device table
// (16-bit fraction of a turn per launch position); later launches of the same grid read it instead of the clock --
// deterministic given the table.  JENGA_ROTATE_TABLE_DUMP=<file> writes the table in front of every replay launch,
// JENGA_ROTATE_TABLE_LOAD=<file> reads it from a file (once per grid size).
void lp_replay_table(long long grid, hipStream_t stream) {
    static unsigned short* table[64] = {nullptr};
    static long long table_n[64] = {0}, loaded_n[64] = {0};
    int dev = 0, mode = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return;
    if (const char* ev = getenv("JENGA_ROTATE_REPLAY")) mode = !strcmp(ev, "record") ? 1 : (!strcmp(ev, "replay") ? 2 : 0);
    if (mode && table_n[dev] < grid) {
        if (table[dev]) (void)hipFree(table[dev]);
        table[dev] = nullptr;
        table_n[dev] = 0;
        if (hipMalloc((void**)&table[dev], (size_t)grid * sizeof(unsigned short)) == hipSuccess &&
            hipMemsetAsync(table[dev], 0, (size_t)grid * sizeof(unsigned short), stream) == hipSuccess)
            table_n[dev] = grid;
        (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_rot_table), &table[dev], sizeof(table[dev]), 0, hipMemcpyHostToDevice,
                                     stream);
    }
    if (mode && !table_n[dev]) mode = 0;
    if (mode == 2 && getenv("JENGA_ROTATE_TABLE_LOAD") && loaded_n[dev] != grid) {
        if (FILE* f = fopen(getenv("JENGA_ROTATE_TABLE_LOAD"), "rb")) {
            std::vector<unsigned short> h((size_t)grid, 0);
            const size_t got = fread(h.data(), sizeof(unsigned short), (size_t)grid, f);
            fclose(f);
            if (got == (size_t)grid && hipStreamSynchronize(stream) == hipSuccess &&
                hipMemcpy(table[dev], h.data(), (size_t)grid * 2, hipMemcpyHostToDevice) == hipSuccess)
                loaded_n[dev] = grid;
        }
        if (loaded_n[dev] != grid) mode = 0;
    }
    if (mode == 2 && getenv("JENGA_ROTATE_TABLE_DUMP")) {
        std::vector<unsigned short> h((size_t)grid, 0);
        if (hipStreamSynchronize(stream) == hipSuccess &&
            hipMemcpy(h.data(), table[dev], (size_t)grid * 2, hipMemcpyDeviceToHost) == hipSuccess)
            if (FILE* f = fopen(getenv("JENGA_ROTATE_TABLE_DUMP"), "wb")) {
                fwrite(h.data(), sizeof(unsigned short), (size_t)grid, f);
                fclose(f);
            }
    }
    (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_rot_table_mode), &mode, sizeof(int), 0, hipMemcpyHostToDevice, stream);
}
#endif

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
    // the launch modes below keep per-device state that is written on the launch stream from host variables or ordered by
    // events: none of that can be recorded into a HIP graph, so a capturing stream gets the plain static launch
    if (flags & (JENGA_ATTN_COHORT | JENGA_ATTN_ROTATE | JENGA_ATTN_BALANCE)) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing((hipStream_t)stream, &cap) != hipSuccess) {
            (void)hipGetLastError();      // (the query itself failed: not this launch's error)
            cap = hipStreamCaptureStatusActive;
        }
        if (cap != hipStreamCaptureStatusNone) flags &= ~(JENGA_ATTN_COHORT | JENGA_ATTN_ROTATE | JENGA_ATTN_BALANCE);
    }
    bool cohort = false;
#if !LP_EXP
    if (flags & JENGA_ATTN_COHORT) {
        set_error("jenga_bsattn_fwd: the cohort start barrier is an experiment (build with JENGA_EXPERIMENTS)");
        return JENGA_EUNSUPPORTED;
    }
#else
    if ((flags & JENGA_ATTN_COHORT) && P.xcd_chunk) {
        // EXPERIMENT: one lazily allocated counter block per device, zeroed on the launch stream in front of every launch,
        // the configuration written to the device global the same way -- launches with this flag must not overlap on one
        // device
        static int* ctr[64] = {nullptr};
        constexpr int STRIDE = 4096;
        int dev = 0;
        (void)hipGetDevice(&dev);
        int size = 64, timeout_us = 300;
        if (const char* ev = getenv("JENGA_COHORT_SIZE")) size = atoi(ev);
        if (const char* ev = getenv("JENGA_COHORT_TIMEOUT_US")) timeout_us = atoi(ev);
        int quorum = size;
        if (const char* ev = getenv("JENGA_COHORT_QUORUM")) quorum = atoi(ev);
        if (quorum < 1 || quorum > size) quorum = size;
        const long long gens = size > 0 ? (B * H * (long long)P.xcd_chunk + size - 1) / size : STRIDE;
        if (dev >= 0 && dev < 64 && size > 0 && gens < STRIDE) {
            if (!ctr[dev] && hipMalloc((void**)&ctr[dev], 8 * STRIDE * sizeof(int)) != hipSuccess) ctr[dev] = nullptr;
            CohortCfg cfg{ctr[dev], size, STRIDE, timeout_us * 100, quorum};
            if (ctr[dev] && hipMemsetAsync(ctr[dev], 0, 8 * STRIDE * sizeof(int), (hipStream_t)stream) == hipSuccess &&
                hipMemcpyToSymbolAsync(HIP_SYMBOL(g_cohort_cfg), &cfg, sizeof(cfg), 0, hipMemcpyHostToDevice,
                                       (hipStream_t)stream) == hipSuccess)
                cohort = true;
        }
    }
#endif
    int rot_ticks = 0;
    if ((flags & JENGA_ATTN_ROTATE) && !cohort) {
        // period of the cursor: JENGA_ROTATE_PERIOD_US=<microseconds>, or (default, "auto") the lifetime of a mid-queue image
        // workgroup of the previous launch with this flag -- copied device to device on the launch stream, so one launch
        // sees one period and no host synchronisation is involved
        int us = 0;
        if (const char* ev = getenv("JENGA_ROTATE_PERIOD_US")) us = atoi(ev);
        rot_ticks = us > 0 ? us * 100 : 1;
        bool auto_period = us <= 0;
#if LP_EXP
        // JENGA_ROTATE_SLOTS=S: position mode with S workgroups resident per XCD (64 = 32 CUs x 2); deterministic
        if (const char* ev = getenv("JENGA_ROTATE_SLOTS")) {
            const int slots = atoi(ev);
            if (slots > 0) rot_ticks = -slots;
            float spread = 0.42f;
            if (const char* es = getenv("JENGA_ROTATE_SPREAD")) spread = (float)atof(es);
            (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_rot_spread), &spread, sizeof(float), 0, hipMemcpyHostToDevice,
                                         (hipStream_t)stream);
        }
        if (rot_ticks < 0) auto_period = false;
        if (rot_ticks > 0) lp_replay_table(grid, (hipStream_t)stream);
#endif
        if (auto_period) {
            void *dst = nullptr, *src = nullptr;
            if (hipGetSymbolAddress(&dst, HIP_SYMBOL(g_rot_period_ticks)) != hipSuccess ||
                hipGetSymbolAddress(&src, HIP_SYMBOL(g_rot_T_est)) != hipSuccess ||
                hipMemcpyAsync(dst, src, sizeof(int), hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
                rot_ticks = 0;
        } else if (rot_ticks != 0 && hipMemcpyToSymbolAsync(HIP_SYMBOL(g_rot_period_ticks), &rot_ticks, sizeof(int), 0,
                                                           hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess) {
            rot_ticks = 0;
        }
    }
    // JENGA_ATTN_BALANCE: query blocks drawn from per-XCD queues, grid oversubscribed by 1/8 (JENGA_BALANCE_EXTRA_PCT) so
    // that a fast XCD has workgroups left to draw from a slow one's queue.  Position mode / replay of the rotated walk index
    // by launch position and stay with the static mapping.
    long long grid_bal = 0;
    int bal_slot = -1;
    bool static_only = (cohort && !(flags & JENGA_ATTN_BALANCE)) || rot_ticks < 0 || (cohort && rot_ticks != 0);
#if LP_EXP
    static_only = static_only || getenv("JENGA_ROTATE_REPLAY") != nullptr;
#endif
    if ((flags & JENGA_ATTN_BALANCE) && P.xcd_chunk && !static_only && (P.n_text_wg_pad & 7) == 0 &&
        (P.img_per_head & 7) == 0) {
        int pct = 12;
        if (const char* ev = getenv("JENGA_BALANCE_EXTRA_PCT")) pct = atoi(ev);
        pct = pct < 0 ? 0 : (pct > 100 ? 100 : pct);
        const long long img = B * H * (long long)P.img_per_head;
        const long long extra = ((img * pct / 100) + 7) & ~7LL;
        if (grid + extra <= 0x7fffffffLL && B * H * (long long)P.xcd_chunk < (1LL << 28)) {
            bal_slot = lp_balance_acquire((hipStream_t)stream);
            if (bal_slot >= 0) {
                grid_bal = grid + extra;
                P.bal_set = bal_slot;
            }
        }
    }
#if LP_EXP
    if (rot_ticks != 0 || grid_bal) lp_times_hook(grid_bal ? grid_bal : grid, (hipStream_t)stream);
#endif
    hipError_t e;
#if LP_EXP
    if (grid_bal && cohort)
        e = dtype == JENGA_BF16 ? lp_launch<BF16, 6>(P, grid_bal, (hipStream_t)stream)
                                : lp_launch<FP16, 6>(P, grid_bal, (hipStream_t)stream);
    else
#endif
    if (grid_bal && rot_ticks > 0)
        e = dtype == JENGA_BF16 ? lp_launch<BF16, 5>(P, grid_bal, (hipStream_t)stream)
                                : lp_launch<FP16, 5>(P, grid_bal, (hipStream_t)stream);
    else if (grid_bal)
        e = dtype == JENGA_BF16 ? lp_launch<BF16, 4>(P, grid_bal, (hipStream_t)stream)
                                : lp_launch<FP16, 4>(P, grid_bal, (hipStream_t)stream);
    else if (rot_ticks != 0)
        e = dtype == JENGA_BF16 ? lp_launch<BF16, 3>(P, grid, (hipStream_t)stream)
                                : lp_launch<FP16, 3>(P, grid, (hipStream_t)stream);
#if LP_EXP
    else if (cohort)
        e = dtype == JENGA_BF16 ? lp_launch<BF16, 1>(P, grid, (hipStream_t)stream)
                                : lp_launch<FP16, 1>(P, grid, (hipStream_t)stream);
#endif
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
