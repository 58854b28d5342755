// Block-sparse attention forward, "LP pair" EXPERIMENT: the LP kernel's pipeline (lp_core.h) in an 8-wave workgroup that
// owns TWO Hilbert-adjacent query blocks of one head -- waves 0-3 query block A = 2j, waves 4-7 block B = 2j+1, two
// waves per SIMD -- and stages every kv block both rows keep ONCE for both (the design the round-1 review asked for).
//
// Lists come from jenga_pair_merge: [shared | A-only | B-only], each ascending.  One tile-step counter tau runs for the
// whole workgroup (one s_barrier per step):
//   dual phase   tau in [0, D), D = 2 max(nA, nB): group g walks ITS OWN unshared list in ITS OWN ring (80 KiB each,
//                the LP ring), staging its own tiles (4 + 4 pieces per wave and step, exactly the LP kernel); the group
//                with the shorter list drains its pipeline and idles (it still meets the barriers);
//   shared phase tau in [D, D + 2 nS): tiles staged ONCE into ring A by all eight waves (2 + 2 pieces per wave and
//                step) and read by both groups; group B's pipeline crosses from ring B to ring A through the runtime
//                tile pointers of the generic lp_bb.  Text blocks and the kv-length mask live at the tail of the shared
//                list (every row keeps the text blocks) and run through lp_slow_tile.
// RESTRICTION of this experiment: no block of an UNSHARED list may need the slow path, i.e. seqlen >= the end of the
// last unshared image block (true for the HunyuanVideo flavour, whose image blocks are never masked); the Python host
// does not route here by default.
#include "../lp_core.h"

namespace jenga {
namespace {

struct Lp2Params {
    const uint16_t* q;
    const uint16_t* k;
    const uint16_t* vt;
    uint16_t* o;
    const int32_t* seqlens;
    const int32_t* pidx;
    const int32_t* pcnt;
    long long q_sb, q_ss, q_sh, k_sb, k_ss, k_sh, o_sb, o_ss, o_sh;
    int B, H, n_blocks, nq_img;
    int npair_img, npair_txt;
    int text_block_start;
    float qk_scale;
    float text_amp;
    int n_text_wg_pad;
    int img_per_head;
    int xcd_chunk;
};

constexpr int LP2_RING = LP_LDS_BYTES;        // bytes of one ring (K 3 slots + V^T 2 slots)
constexpr int LP2_LDS_BYTES = 2 * LP2_RING;   // 160 KiB: one workgroup per CU

__device__ __forceinline__ void lp_stage2(const void* base, unsigned lds, unsigned o0, unsigned o1) {
#ifdef JENGA_X_NODMA
    return;
#endif
    asm volatile("s_mov_b32 m0, %0\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %2, %1\n\t"
                 "global_load_lds_dwordx4 %3, %1 offset:1024"
                 :
                 : "s"(lds), "s"(base), "v"(o0), "v"(o1)
                 : "memory", "m0");
}

template <typename T, bool TEXT>
__device__ __forceinline__ void attn_pair_lp(const Lp2Params& P, unsigned char* smem, int b, int h, int pr) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave8 >> 2, wave_u = wave8 & 3;
    const int lq = lane & 31, hi = lane >> 5;
    const int seqlen = P.seqlens ? __builtin_amdgcn_readfirstlane(P.seqlens[b]) : P.n_blocks * 128;

    // ---- lists ----
    const int32_t* plist = nullptr;
    int ns, na, nb;
    if (TEXT) {
        ns = P.n_blocks;
        na = nb = 0;
    } else {
        const long long prow = ((long long)b * P.H + h) * P.npair_img + pr;
        plist = P.pidx + prow * P.n_blocks;
        ns = __builtin_amdgcn_readfirstlane(P.pcnt[prow * 4 + 0]);
        na = __builtin_amdgcn_readfirstlane(P.pcnt[prow * 4 + 1]);
        nb = __builtin_amdgcn_readfirstlane(P.pcnt[prow * 4 + 2]);
        if (2 * pr + 1 >= P.nq_img) {   // last pair of an odd count: one row, everything it keeps is "A-only" (ns = 0)
            ns = na;                    // -> walk it as the shared list (same offset 0), staged by all eight waves
            na = 0;
        }
    }
    const int n_own = g ? nb : na;
    const int own_off = ns + (g ? na : 0);
    const int D = 2 * (na > nb ? na : nb);

    // ---- this group's query block ----
    const int m_raw = (TEXT ? P.nq_img : 0) + 2 * pr + g;
    const bool has_q = TEXT ? (m_raw < P.n_blocks) : (m_raw < P.nq_img);
    const int m = has_q ? m_raw : m_raw - 1;   // (a missing B block: loads stay in range, nothing is computed or stored)

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

    const unsigned char* kbh = reinterpret_cast<const unsigned char*>(P.k + b * P.k_sb + h * P.k_sh);
    const unsigned char* vbh =
        reinterpret_cast<const unsigned char*>(P.vt + ((long long)b * P.H + h) * (long long)P.n_blocks * (2 * 128 * 64));

    int k_addr[8], v_addr[4];
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) k_addr[ds] = LP_K_RING + lq * 256 + (((ds * 2 + hi) ^ (lq & 15)) << 4);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        v_addr[ks] = LP_V_RING + lq * 128 + ((((ks >> 1) * 4 + hi * 2 + (ks & 1)) ^ ((lq >> 1) & 7)) << 4);
    const unsigned smem_base =
        __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem);
    const unsigned kss_b = (unsigned)P.k_ss * 2u;
    // own tiles (dual phase): this wave stages a quarter of a tile, 4 pieces (the LP kernel's offsets)
    const int kc_ = lane & 15, ksw_ = lane >> 4;
    const int kr_ = 16 * wave_u + ksw_;
    const unsigned k_src0 = (unsigned)(kr_ + 0) * kss_b + ((kc_ ^ (0 + ksw_)) << 4);
    const unsigned k_src1 = (unsigned)(kr_ + 4) * kss_b + ((kc_ ^ (4 + ksw_)) << 4) - 1024u;
    const unsigned k_src2 = (unsigned)(kr_ + 8) * kss_b + ((kc_ ^ (8 + ksw_)) << 4) - 2048u;
    const unsigned k_src3 = (unsigned)(kr_ + 12) * kss_b + ((kc_ ^ (12 + ksw_)) << 4) - 3072u;
    const int vc_ = lane & 7, vsw_ = lane >> 4;
    const int vr_ = 32 * wave_u + (lane >> 3);
    const unsigned v_src0 = (unsigned)(vr_ + 0) * 128 + ((vc_ ^ ((0 + vsw_) & 7)) << 4);
    const unsigned v_src1 = (unsigned)(vr_ + 8) * 128 + ((vc_ ^ ((4 + vsw_) & 7)) << 4) - 1024u;
    const unsigned v_src2 = (unsigned)(vr_ + 16) * 128 + ((vc_ ^ ((8 + vsw_) & 7)) << 4) - 2048u;
    const unsigned v_src3 = (unsigned)(vr_ + 24) * 128 + ((vc_ ^ ((12 + vsw_) & 7)) << 4) - 3072u;
    // shared tiles: this wave stages an eighth of a tile, 2 pieces: K rows 8 wave8 + ksw + {0, 4} (swizzle = row & 15),
    // V^T rows 16 wave8 + (lane >> 3) + {0, 8} (swizzle = (row >> 1) & 7)
    const int kr2_ = 8 * wave8 + ksw_, ksz_ = 8 * (wave8 & 1) + ksw_;
    const unsigned k2_src0 = (unsigned)(kr2_ + 0) * kss_b + ((kc_ ^ ((ksz_ + 0) & 15)) << 4);
    const unsigned k2_src1 = (unsigned)(kr2_ + 4) * kss_b + ((kc_ ^ ((ksz_ + 4) & 15)) << 4) - 1024u;
    const int vr2_ = 16 * wave8 + (lane >> 3);
    const unsigned v2_src0 = (unsigned)(vr2_ + 0) * 128 + ((vc_ ^ ((0 + vsw_) & 7)) << 4);
    const unsigned v2_src1 = (unsigned)(vr2_ + 8) * 128 + ((vc_ ^ ((4 + vsw_) & 7)) << 4) - 1024u;

    // list windows: 64 entries of the own list and 64 of the shared list, one VGPR each (bsattn.hip)
    int oc = 0, ob = -64, sc = 0, sb = -64;
    auto own_at = [&](int i) -> int {
        if (i < ob || i >= ob + 64) {
            ob = i & ~63;
            oc = (ob + lane < n_own) ? plist[own_off + ob + lane] : 0;
            __builtin_amdgcn_s_waitcnt(0x0F70);
        }
        return __builtin_amdgcn_readlane(oc, i - ob);
    };
    auto sh_at = [&](int i) -> int {
        if (i >= ns) i = ns - 1;
        if (TEXT) return i;
        if (i < sb || i >= sb + 64) {
            sb = i & ~63;
            sc = (sb + lane < ns) ? plist[sb + lane] : 0;
            __builtin_amdgcn_s_waitcnt(0x0F70);
        }
        return __builtin_amdgcn_readlane(sc, i - sb);
    };

    // shared tiles that need the slow path (tail of the ascending shared list), tiles entirely behind the kv length
    int n_fast_sh = ns;
    if (!TEXT) {
        while (n_fast_sh > 0) {
            const int bl = sh_at(n_fast_sh - 1);
            if (bl >= P.text_block_start || (bl + 1) * 128 > seqlen) --n_fast_sh; else break;
        }
    }
    int T_all = D + 2 * ns;
    if (!TEXT) {
        while (T_all > D && sh_at((T_all - 1 - D) >> 1) * 128 + ((T_all - 1 - D) & 1) * 64 >= seqlen) --T_all;
    }
    const int T_fast = D + 2 * n_fast_sh < T_all ? D + 2 * n_fast_sh : T_all;

    // tile tau: ring, activity, staging.  Returns the number of LDS-DMA pieces this wave issued.
    auto is_act = [&](int t) -> bool { return has_q && t >= 0 && (t < D ? (t >> 1) < n_own : t < T_all); };
    auto ring_of = [&](int t) -> int { return t < D ? g : 0; };
    auto kptr = [&](int t) { return smem + ring_of(t) * LP2_RING + (t % 3) * LP_TILE; };
    auto vptr = [&](int t) { return smem + ring_of(t) * LP2_RING + (t & 1) * LP_TILE; };
    auto stage_k = [&](int t) -> int {
        if (t >= T_all) return 0;
        if (t < D) {
            if ((t >> 1) >= n_own) return 0;
            const int blk = own_at(t >> 1);
            lp_stage4(kbh + (unsigned long long)((unsigned)blk * 128u + (unsigned)(t & 1) * 64u) * kss_b,
                      smem_base + g * LP2_RING + LP_K_RING + (t % 3) * LP_TILE + wave_u * 4096, k_src0, k_src1, k_src2,
                      k_src3);
            return 4;
        }
        const int blk = sh_at((t - D) >> 1);
        lp_stage2(kbh + (unsigned long long)((unsigned)blk * 128u + (unsigned)((t - D) & 1) * 64u) * kss_b,
                  smem_base + LP_K_RING + (t % 3) * LP_TILE + wave8 * 2048, k2_src0, k2_src1);
        return 2;
    };
    auto stage_v = [&](int t) -> int {
        if (t >= T_all) return 0;
        if (t < D) {
            if ((t >> 1) >= n_own) return 0;
            const int blk = own_at(t >> 1);
            lp_stage4(vbh + (unsigned long long)((unsigned)blk * 2u + (unsigned)(t & 1)) * (128u * 64u * 2u),
                      smem_base + g * LP2_RING + LP_V_RING + (t & 1) * LP_TILE + wave_u * 4096, v_src0, v_src1, v_src2,
                      v_src3);
            return 4;
        }
        const int blk = sh_at((t - D) >> 1);
        lp_stage2(vbh + (unsigned long long)((unsigned)blk * 2u + (unsigned)((t - D) & 1)) * (128u * 64u * 2u),
                  smem_base + LP_V_RING + (t & 1) * LP_TILE + wave8 * 2048, v2_src0, v2_src1);
        return 2;
    };
    auto wait_keep = [&](int n) {   // everything but the youngest n pieces of this wave has landed
        if (n >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    f32x16 sA, sB;
    uint4 pfA[2], pfB[2];
    uint4 frk[8];
#pragma unroll
    for (int r = 0; r < 16; ++r) sA[r] = sB[r] = 0.f;
    pfA[0] = pfA[1] = pfB[0] = pfB[1] = make_uint4(0u, 0u, 0u, 0u);

    // prologue: K(0), K(1)
    stage_k(0);
    stage_k(1);
    LP_WAIT_ALL();
    __syncthreads();

    // ---- fast steps: stage V^T(t) and K(t+2); QK^T on K(t); P.V on V^T(t-1) ----
    // Three ranges of tau with a constant role per group -- [0, 2 n_own) active, [2 n_own, D) idle, [D, T_fast) active --
    // so that the hot loops hold the steady form only; fill (pipeline empty) and drain happen at range starts.
    bool pending = false;   // an item is in the pipeline (wave-uniform)
    const int own_end = has_q ? (2 * n_own < T_fast ? 2 * n_own : T_fast) : 0;
    const int dual_end = D < T_fast ? D : T_fast;
#define LP2_STEP(T_, PV_, SM0_)                                                                                       \
    do {                                                                                                              \
        const unsigned char* kt_ = kptr(T_);                                                                          \
        const unsigned char* vt_ = vptr((T_) > 0 ? (T_) - 1 : 0);                                                     \
        stage_v(T_);                                                                                                  \
        lp_bb<T, TEXT, 0, PV_, true, SM0_>(st, kt_, vt_, sA, sB, pfA, pfB, k_addr, v_addr, P.qk_scale, frk);          \
        const int nk_ = stage_k((T_) + 2);                                                                            \
        lp_bb<T, TEXT, 1, PV_, true, true>(st, kt_, vt_, sB, sA, pfB, pfA, k_addr, v_addr, P.qk_scale, frk);          \
        wait_keep(nk_);                                                                                               \
        __syncthreads();                                                                                              \
    } while (0)
    for (int rg = 0; rg < 3; ++rg) {
        const int t0 = rg == 0 ? 0 : rg == 1 ? own_end : dual_end;
        const int t1 = rg == 0 ? own_end : rg == 1 ? dual_end : T_fast;
        const bool active = has_q && rg != 1;
        if (t0 >= t1) continue;
        if (active) {
            int t = t0;
            if (!pending) {
                LP2_STEP(t, false, false);
                ++t;
            }
            for (; t < t1; ++t) LP2_STEP(t, true, true);
            pending = true;
        } else {
            for (int t = t0; t < t1; ++t) {
                stage_v(t);
                if (pending) {   // drain inside the first idle step: its V^T tile (t - 1) is still in the ring
                    const unsigned char* vt_ = vptr(t - 1);
                    lp_bb<T, TEXT, 0, true, false, true>(st, nullptr, vt_, sA, sB, pfA, pfB, k_addr, v_addr, P.qk_scale,
                                                         frk);
                    lp_bb<T, TEXT, 1, true, false, false>(st, nullptr, vt_, sB, sA, pfB, pfA, k_addr, v_addr,
                                                          P.qk_scale, frk);
                    pending = false;
                }
                const int nk = stage_k(t + 2);
                wait_keep(nk);
                __syncthreads();
            }
        }
    }
#undef LP2_STEP
    // drain: softmax of the last item, P.V of the last tile
    if (pending) {
        const unsigned char* vt_ = vptr(T_fast - 1);
        lp_bb<T, TEXT, 0, true, false, true>(st, nullptr, vt_, sA, sB, pfA, pfB, k_addr, v_addr, P.qk_scale, frk);
        lp_bb<T, TEXT, 1, true, false, false>(st, nullptr, vt_, sB, sA, pfB, pfA, k_addr, v_addr, P.qk_scale, frk);
    }
    // ---- slow tiles (shared list tail: text blocks, kv-length mask) ----
    if (!TEXT) {
        for (int t = T_fast; t < T_all; ++t) {
            stage_v(t);
            stage_k(t + 2);
            LP_WAIT_ALL();   // this step reads V^T(t) itself
            __syncthreads();
            if (has_q) {
                const int blk = sh_at((t - D) >> 1);
                lp_slow_tile<T>(st, kptr(t), vptr(t), blk * 128 + ((t - D) & 1) * 64, blk >= P.text_block_start,
                                P.text_amp, seqlen, hi, k_addr, v_addr);
            }
            __syncthreads();
        }
    }
    LP_WAIT_ALL();

    // ---- epilogue: o = acc / l, rows >= seqlen written as zeros (image rows only) ----
    if (has_q) {
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
}

template <typename T>
__global__ void __launch_bounds__(512) bsattn_lp2_kernel(Lp2Params P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int id = blockIdx.x;
    if (id < P.n_text_wg_pad) {   // text query block pairs first: the longest work items start earliest
        if (id >= P.B * P.H * P.npair_txt) return;
        const int bh = id / P.npair_txt;
        attn_pair_lp<T, true>(P, smem, bh / P.H, bh % P.H, id % P.npair_txt);
        return;
    }
    const int li = id - P.n_text_wg_pad;
    const int bh = li / P.img_per_head;
    const int r = li % P.img_per_head;
    int pr;
    if (P.xcd_chunk) {
        pr = (r & 7) * P.xcd_chunk + (r >> 3);
        if ((r >> 3) >= P.xcd_chunk || pr >= P.npair_img) return;
    } else {
        pr = r;
    }
    attn_pair_lp<T, false>(P, smem, bh / P.H, bh % P.H, pr);
}

}  // namespace
}  // namespace jenga

using namespace jenga;

// same arguments as jenga_bsattn_pair_fwd (bsattn2.hip), already validated there; reached through it with JENGA_ATTN_LP
int jenga_bsattn_lp2_launch(void* stream, const void* q, const void* k, const void* vt, void* o, const int32_t* seqlens,
                            const int32_t* pidx, const int32_t* pcnt, int64_t B, int64_t H, int64_t n_blocks,
                            int64_t nq_img, int64_t q_sb, int64_t q_ss, int64_t q_sh, int64_t k_sb, int64_t k_ss,
                            int64_t k_sh, int64_t o_sb, int64_t o_ss, int64_t o_sh, float sm_scale, float text_amp,
                            int64_t text_block_start, int dtype, int flags) {
    Lp2Params P;
    P.q = (const uint16_t*)q;
    P.k = (const uint16_t*)k;
    P.vt = (const uint16_t*)vt;
    P.o = (uint16_t*)o;
    P.seqlens = seqlens;
    P.pidx = pidx;
    P.pcnt = pcnt;
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
    const size_t smem = LP2_LDS_BYTES;
    if (dtype == JENGA_BF16) {
        (void)hipFuncSetAttribute((const void*)bsattn_lp2_kernel<BF16>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)smem);
        hipLaunchKernelGGL(bsattn_lp2_kernel<BF16>, dim3((unsigned)grid), dim3(512), smem, (hipStream_t)stream, P);
    } else {
        (void)hipFuncSetAttribute((const void*)bsattn_lp2_kernel<FP16>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)smem);
        hipLaunchKernelGGL(bsattn_lp2_kernel<FP16>, dim3((unsigned)grid), dim3(512), smem, (hipStream_t)stream, P);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("jenga_bsattn_pair_fwd (lp pair): %s", hipGetErrorString(e));
        return JENGA_ELAUNCH;
    }
    return JENGA_OK;
}
