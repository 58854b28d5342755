// Block selection (K3-5).  One workgroup = 4 consecutive query blocks of one (batch, head).
//   phase A (the workgroup): the pooled scores of the 4 rows.  The head's pooled K goes through LDS in tiles of 256 columns x
//     64 channels with coalesced global loads two tiles ahead; a thread owns one column of the tile and runs, per (row,
//     column), the same 128 sequential fused multiply-adds as ever, the pooled Q of the 4 rows in SGPRs (the K value is
//     unpacked once for the 4 rows).
//   phase B: ONE WAVE OWNS ONE ROW and runs softmax -> sort -> kept-count rule -> bit set -> compaction without another
//     workgroup barrier: the sort is a bitonic network over 16 keys per lane (registers + wave shuffles), the cumulative sum
//     an exact shuffle scan, the prefix popcount a shuffle scan.
// With (batch * heads) a multiple of 8 all workgroups of a head run on ONE XCD (id % 8): its pooled K (230 KB) stays in that
// XCD's L2.
// History at the 720p shape, 24 heads x 900 x 902, flat / coherent lists (profiles/r05_select_ab.json):
//   round 4   one row per workgroup, one-thread cumulative sum                                    0.69 / 0.58 ms
//   round 5a  2 rows per workgroup run one after the other by all 4 waves (sort: 6 barriers per row),
//             exact shuffle scan, (b, h) -> XCD map                                                0.63 / 0.50
//   round 5b  wave per row, 4 rows per K value, K row read by its own thread (64 lines per load
//             instruction: the texture path set the pace, 0.225 of the 0.37)                       0.37 / 0.38
//   round 5c  K through LDS tiles, coalesced, one tile ahead                                       0.33 / 0.33
//   round 5d  two tiles ahead (111 VGPRs = the 4 waves per SIMD the 32 KB tile allows anyway)      0.28 / 0.29
//   round 5e  scalar fmas instead of v_pk_fma_f32 (see build.py), no SLP vectorisation            0.28 / 0.28
//             (passes of loads in flight, same box: 1 -> 0.291 / 0.295, 2 -> 0.281 / 0.282, 3 -> 0.312 / 0.318: 143 VGPRs)
//   (8 rows per workgroup, two per wave: 0.40 -- 132 VGPRs, 3 waves per SIMD.)
// What is left (counter passes, profiles/r05_pmc_select.json): 6.2 k VALU instructions per row-wave -- ~2.6 k dot products,
// ~1.9 k sort, ~0.5 k softmax -- keep the VALU >= 84 % busy: the kernel is instruction-bound now.
// The per-row work:
//   scores -> softmax (dtype) -> sort -> cumulative-probability / top-k rule -> bit set
//   -> OR static neighbours / first-frame / text columns -> ascending index list (+ optional one-hot mask).
// Rounding points follow the reference's torch code running in the tensor dtype
// (attention_block_triton_diffres.py:221-250); see include/jenga_amd.h.
#include <cstring>

#include "common.h"

namespace jenga {
namespace {

typedef float sel_f2 __attribute__((ext_vector_type(2)));

constexpr int MAX_PER_THREAD = 8;  // 256 threads x 8 = up to 2048 image key blocks
#ifndef SEL_ROWS
#define SEL_ROWS 4
#endif
constexpr int SEL_R = SEL_ROWS;    // query blocks (rows) per workgroup; wave w owns rows w, w + 4, ...: a multiple of 4
// elimination switches (A/B builds, wrong results): SEL_X & 1 no dot products, 2 no sort, 4 no cumulative sum, 8 no output
#ifndef SEL_X
#define SEL_X 0
#endif

// LDS traffic between the lanes of ONE wave (a wave's DS operations execute in order; this only keeps the compiler from
// moving them across the point)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Bitonic sort (descending) of 64 * E unique 32-bit keys held by ONE wave, E consecutive keys per lane in registers
// (element i = lane * E + r): compare-exchange distances below E stay inside a lane, every larger distance is a wave
// shuffle -- no LDS, no barrier.  The network is the "mirror" form: every merge of two sorted k/2-runs starts by comparing
// element i with i ^ (k - 1) (the second run read backwards), then i with i ^ j for j = k/4 .. 1; the lower index always
// keeps the larger key, so an exchange inside a lane is one v_max + one v_min with compile-time registers, and one across
// lanes is v_max + v_min + a select on a lane-bit mask.  Keys are unique, so every correct sort gives the same order.
template <int E>
__device__ __forceinline__ void wave_bitonic_sort_desc(uint32_t (&v)[E], int lane) {
#pragma unroll
    for (int k = 2; k <= 64 * E; k <<= 1) {
        if (k <= E) {
#pragma unroll
            for (int r = 0; r < E; ++r) {
                const int pr = r ^ (k - 1);
                if (r < pr) {
                    const uint32_t a = v[r], b = v[pr];
                    v[r] = a > b ? a : b;
                    v[pr] = a > b ? b : a;
                }
            }
        } else {
            const bool take_max = ((lane * E) & (k >> 1)) == 0;
            uint32_t pv[E];
#pragma unroll
            for (int r = 0; r < E; ++r) pv[r] = (uint32_t)__shfl_xor((int)v[E - 1 - r], k / E - 1);
#pragma unroll
            for (int r = 0; r < E; ++r) {
                const uint32_t mx = v[r] > pv[r] ? v[r] : pv[r], mn = v[r] > pv[r] ? pv[r] : v[r];
                v[r] = take_max ? mx : mn;
            }
        }
#pragma unroll
        for (int j = k >> 2; j >= 1; j >>= 1) {
            if (j >= E) {
                const bool take_max = ((lane * E) & j) == 0;
                uint32_t pv[E];      // all E shuffles in flight before the first use
#pragma unroll
                for (int r = 0; r < E; ++r) pv[r] = (uint32_t)__shfl_xor((int)v[r], j / E);
#pragma unroll
                for (int r = 0; r < E; ++r) {
                    const uint32_t mx = v[r] > pv[r] ? v[r] : pv[r], mn = v[r] > pv[r] ? pv[r] : v[r];
                    v[r] = take_max ? mx : mn;
                }
            } else {
#pragma unroll
                for (int r = 0; r < E; ++r) {
                    if ((r & j) == 0) {
                        const uint32_t a = v[r], b = v[r | j];
                        v[r] = a > b ? a : b;
                        v[r | j] = a > b ? b : a;
                    }
                }
            }
        }
    }
}

// E = keys per lane; the row is padded to 64 * E >= nk_img keys.
template <typename T, int E>
__global__ void __launch_bounds__(256)
block_select_kernel(const uint16_t* __restrict__ qpool, const uint16_t* __restrict__ kpool,
                    const uint8_t* __restrict__ neighbors, int nb_rows, int nb_cols, uint8_t* __restrict__ mask,
                    int32_t* __restrict__ idx, int32_t* __restrict__ cnt, int BH, int nq, int nk_img, int text_blocks,
                    int top_k, float p_thr, int first_frame_blocks, int scan_log_nx, int ngrp, int xcd_map, float scale) {
    constexpr int NP = 64 * E;                                           // padded row length
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BUF_WORDS = SEL_R * NP > 8192 ? SEL_R * NP : 8192;     // phase A's K tile (32 KB), then the row buffers
    uint32_t* rowbuf = reinterpret_cast<uint32_t*>(smem);                // [SEL_R][NP]: scores (float), then sorted keys
    uint32_t* bits_all = rowbuf + BUF_WORDS;                             // [SEL_R][80]: up to 2048 + 512 columns
    uint32_t* wpre_all = bits_all + SEL_R * 80;                          // [SEL_R][80] exclusive popcount prefix
    float* sbuf_all = reinterpret_cast<float*>(wpre_all + SEL_R * 80);   // [SEL_R][2 << scan_log_nx] (device-scan, W > 64)

    const int nk_all = nk_img + text_blocks;
    const int tid = threadIdx.x;
    // workgroup id -> (bh, row group).  xcd_map: the hardware deals ids to the XCDs round robin (id % 8); (b, h) pair bh
    // runs on XCD bh % 8, so that its pooled K stays in ONE L2.  (Only when BH % 8 == 0: a rank that holds 3 heads must
    // not leave 5 XCDs idle.)
    long long bh;
    int grp;
    if (xcd_map) {
        const int x = blockIdx.x & 7;
        const long long k_ = blockIdx.x >> 3;
        bh = x + 8 * (k_ / ngrp);
        grp = (int)(k_ % ngrp);
        if (bh >= BH) return;
    } else {
        bh = blockIdx.x / ngrp;
        grp = blockIdx.x % ngrp;
    }
    const int m0 = grp * SEL_R;
    const int g_n = nq - m0 < SEL_R ? nq - m0 : SEL_R;      // rows of this group

    // ---- phase A: pooled scores for the image columns (K3) ----
    // The head's pooled K goes through LDS in tiles of 256 columns x 64 channels (one 128-byte line per row): global loads
    // with 8 consecutive lanes on one line (a thread reading its own 256-byte row touched 64 lines per load instruction and
    // the texture path, not the FMAs, set the pace: 0.225 of 0.37 ms), the 16-byte chunk c of row r stored at slot
    // c ^ ((r >> 1) & 7) so that both the stores (8 lanes per row) and the loads (one row per lane) are conflict-free.  The
    // global loads of the next TWO passes are in flight while this one is consumed (one pass ahead left 0.05 ms exposed).  The
    // scores stay in registers until the last tile is done; then the tile buffer becomes the row buffer.
    {
        uint4* tile = reinterpret_cast<uint4*>(smem);          // [256 rows][8 slots]
        // rows past the end of the head repeat the last one (read, never used)
        const uint4* qr[SEL_R];
#pragma unroll
        for (int g = 0; g < SEL_R; ++g)
            qr[g] = reinterpret_cast<const uint4*>(qpool + (bh * nq + m0 + (g < g_n ? g : g_n - 1)) * 128);
        const uint4* kbase = reinterpret_cast<const uint4*>(kpool + bh * nk_all * 128);     // 16 chunks per row
        const int ntile = (nk_img + 255) >> 8;
        const int lr = tid >> 3, lc = tid & 7;
#ifndef SEL_AHEAD
#define SEL_AHEAD 2
#endif
        uint4 pre[SEL_AHEAD][8];        // pass t lives in pre[t % SEL_AHEAD]: that many passes of global loads in flight
        auto gload = [&](uint4 (&dst)[8], int t) {       // pass t = (column tile t >> 1, channel half t & 1)
            const int r0 = (t >> 1) * 256 + lr;
#pragma unroll
            for (int s_ = 0; s_ < 8; ++s_) {
                const int r = r0 + s_ * 32;
                dst[s_] = (r < nk_img && !(SEL_X & 1)) ? kbase[(long long)r * 16 + (t & 1) * 8 + lc] : make_uint4(0u, 0u, 0u, 0u);
            }
        };
        float scv[MAX_PER_THREAD][SEL_R];
#pragma unroll
        for (int t = 0; t < SEL_AHEAD; ++t)
            if (t < 2 * ntile) gload(pre[t], t);
#pragma unroll
        for (int ct = 0; ct < MAX_PER_THREAD; ++ct) {
            if (ct * 256 < NP && ct < ntile) {
                sel_f2 acc[SEL_R / 2];
#pragma unroll
                for (int g2 = 0; g2 < SEL_R / 2; ++g2) acc[g2] = (sel_f2){0.f, 0.f};
#pragma unroll
                for (int half = 0; half < 2; ++half) {
#pragma unroll
                    for (int s_ = 0; s_ < 8; ++s_) {
                        const int r = s_ * 32 + lr;
                        tile[r * 8 + (lc ^ ((r >> 1) & 7))] = pre[(ct * 2 + half) % SEL_AHEAD][s_];
                    }
                    __syncthreads();
                    if (ct * 2 + half + SEL_AHEAD < 2 * ntile) gload(pre[(ct * 2 + half) % SEL_AHEAD], ct * 2 + half + SEL_AHEAD);
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        float f[8];
                        unpack8<T>(tile[tid * 8 + (c ^ ((tid >> 1) & 7))], f);
                        float qf[SEL_R][8];
#pragma unroll
                        for (int g = 0; g < SEL_R; ++g) unpack8<T>(qr[g][half * 8 + c], qf[g]);
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
#ifdef SEL_PK           // A/B: two rows per v_pk_fma_f32 (0.316 / 0.324 ms against 0.278 / 0.278 with the scalar chains: the
                        // phase is latency-bound, and packed fp32 arithmetic is avoided in this library, see build.py)
                            const sel_f2 ff = (sel_f2){f[e], f[e]};
#pragma unroll
                            for (int g2 = 0; g2 < SEL_R / 2; ++g2)
                                acc[g2] = __builtin_elementwise_fma((sel_f2){qf[2 * g2][e], qf[2 * g2 + 1][e]}, ff, acc[g2]);
#else
#pragma unroll
                            for (int g2 = 0; g2 < SEL_R / 2; ++g2) {
                                acc[g2].x = fmaf(qf[2 * g2][e], f[e], acc[g2].x);
                                acc[g2].y = fmaf(qf[2 * g2 + 1][e], f[e], acc[g2].y);
                            }
#endif
                        }
                    }
                    __syncthreads();      // the tile is overwritten by the next pass (or by the scores)
                }
#pragma unroll
                for (int g2 = 0; g2 < SEL_R / 2; ++g2) {
                    if (SEL_X & 1) {
                        const int j = ct * 256 + tid;
                        acc[g2] = (sel_f2){(float)((j * 37 + g2 * 22) % 97) * 0.01f, (float)((j * 37 + g2 * 22 + 11) % 97) * 0.01f};
                    }
                    scv[ct][2 * g2] = round_to<T>(round_to<T>(acc[g2].x) * scale);
                    scv[ct][2 * g2 + 1] = round_to<T>(round_to<T>(acc[g2].y) * scale);
                }
            }
        }
        float* scw = reinterpret_cast<float*>(rowbuf);
#pragma unroll
        for (int ct = 0; ct < MAX_PER_THREAD; ++ct) {
            const int j = ct * 256 + tid;
            if (ct * 256 < NP && j < nk_img) {
#pragma unroll
                for (int g = 0; g < SEL_R; ++g) scw[g * NP + j] = scv[ct][g];
            }
        }
    }
    __syncthreads();

    // ---- phase B: wave g on row m0 + g ----
    const int lane = tid & 63;
#pragma unroll 1
    for (int g = tid >> 6; g < g_n; g += 4) {
    const int m = m0 + g;
    const long long row = bh * nq + m;
    uint32_t* keys = rowbuf + g * NP;
    uint32_t* bits = bits_all + g * 80;
    uint32_t* wpre = wpre_all + g * 80;
    bits[lane] = 0u;
    if (lane < 16) bits[64 + lane] = 0u;

    // softmax over the image columns, rounded to dtype (K4); lane holds columns lane * E .. lane * E + E - 1
    float sc[E];
    float lmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const int j = lane * E + r;
        sc[r] = j < nk_img ? __uint_as_float(keys[j]) : -INFINITY;
        lmax = fmaxf(lmax, sc[r]);
    }
    const float rmax = wave_max(lmax);
    float lsum = 0.f;
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const int j = lane * E + r;
        if (j < nk_img) {
            sc[r] = expf(sc[r] - rmax);
            lsum += sc[r];
        }
    }
    const float rsum = wave_sum(lsum);
    uint32_t v[E];
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const int j = lane * E + r;
        v[r] = j < nk_img ? (((uint32_t)from_f32<T>(sc[r] / rsum) << 16) | (uint32_t)(0xFFFF - j)) : 0u;  // padding sorts last
    }
    // sort, descending on (probability bits, then lower column first)
    if (!(SEL_X & 2)) wave_bitonic_sort_desc<E>(v, lane);
    wave_sync();        // every lane has its scores in registers: the row buffer now takes the sorted keys
#pragma unroll
    for (int r = 0; r < E; ++r) keys[lane * E + r] = v[r];
    wave_sync();

    // ---- n = max(#(cumsum <= p) + 1, top_k) ----
    int count = 0;
    if (SEL_X & 4) {
        count = top_k - 1;
    } else if (scan_log_nx < 0) {
        // default contract = torch.cumsum of a 16-bit tensor on the CPU (what the reference's goldens were generated
        // with): sequential fp32 accumulation, each partial rounded to dtype.  The wave takes 64 sorted probabilities at a
        // time (one LDS read per lane).
        //  * Fast path, bit-identical by construction: the probabilities are 16-bit values and the partial sums stay below
        //    2, so as long as every addend so far is at least 2^-16 (bf16; 2^-13 for fp16) every partial sum -- in ANY order
        //    of addition -- is a multiple of 2^-24 below 2 and therefore exact in fp32: a shuffle scan gives the sequential
        //    chain's values.  The list is sorted descending, so one look at the chunk's last element decides.
        //  * Otherwise (tiny probabilities before the threshold is reached: p near 1, very peaky rows) the chunk runs the
        //    sequential chain itself, every lane redundantly on values broadcast with v_readlane.
        // The partial sums are non-decreasing, so "count while round(acc) <= p, stop at the first miss" equals "count every
        // i with round(acc_i) <= p"; a chunk whose last partial already missed ends the walk.
        const float exact_min = __is_same(T, BF16) ? 1.52587890625e-05f : 1.220703125e-04f;
        float acc = 0.f;
        for (int c0 = 0; c0 < nk_img; c0 += 64) {
            const int col = c0 + lane;
            const uint32_t mine = col < nk_img ? (keys[col] >> 16) : 0u;
            const float pv = to_f32<T>((uint16_t)mine);
            const int last_col = c0 + 63 < nk_img ? 63 : nk_img - 1 - c0;
            const float smallest = __shfl(pv, last_col);
            if (smallest >= exact_min) {
                float inc = pv;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const float t_ = __shfl_up(inc, o);
                    if (lane >= o) inc += t_;
                }
                const float part = acc + inc;
                count += __popcll(__builtin_amdgcn_ballot_w64(col < nk_img && round_to<T>(part) <= p_thr));
                acc = __shfl(part, 63);
            } else {
#pragma unroll
                for (int j = 0; j < 64; ++j) {
                    const float pj = to_f32<T>((uint16_t)__builtin_amdgcn_readlane((int)mine, j));
                    acc = acc + pj;
                    count += (c0 + j < nk_img && round_to<T>(acc) <= p_thr) ? 1 : 0;
                }
            }
            if (!(round_to<T>(acc) <= p_thr)) break;      // wave-uniform: every lane holds the same value
        }
    } else {
        // JENGA_SELECT_DEVICE_SCAN: torch.cumsum of a 16-bit tensor on the DEVICE, restated (ATen/native/cuda/
        // ScanUtils.cuh, tensor_kernel_scan_innermost_dim_impl, torch 2.10): the row is scanned in chunks of
        // W = 2 * nx columns (nx = 2^scan_log_nx threads, chosen by get_log_num_threads_x_inner_scan from the number
        // of rows and the row length); inside a chunk a Sklansky network, EVERY add rounded to the 16-bit dtype
        // (row_buf is scalar_t); the chunk's last element (also 16-bit) is added to the next chunk's first one.
        // Partials are not monotonic, so ALL columns with cumsum <= p are counted ((cumsum <= p).sum(), :244-245).
        const int nx = 1 << scan_log_nx, W = 2 * nx;
        if (W <= 64) {
            float total = 0.f;      // the chunk in registers, Sklansky steps as wave shuffles
            for (int c0 = 0; c0 < nk_img; c0 += W) {
                const int col = c0 + lane;
                float x = (lane < W && col < nk_img) ? to_f32<T>((uint16_t)(keys[col] >> 16)) : 0.f;
                if (lane == 0) x = round_to<T>(x + total);
                for (int s_ = 0; s_ <= scan_log_nx; ++s_) {
                    const int sft = 1 << s_;
                    const int src = (lane & ~(2 * sft - 1)) + sft - 1;
                    const float o = __shfl(x, src & 63);
                    if (lane & sft) x = round_to<T>(x + o);
                }
                if (lane < W && col < nk_img && x <= p_thr) ++count;
                total = __shfl(x, W - 1);
            }
        } else {
            float* sbuf = sbuf_all + (size_t)g * W;
            float total = 0.f;
            for (int c0 = 0; c0 < nk_img; c0 += W) {
                for (int e = lane; e < W; e += 64)
                    sbuf[e] = (c0 + e < nk_img) ? to_f32<T>((uint16_t)(keys[c0 + e] >> 16)) : 0.f;
                wave_sync();
                if (lane == 0) sbuf[0] = round_to<T>(sbuf[0] + total);
                wave_sync();
                for (int s_ = 0; s_ <= scan_log_nx; ++s_) {
                    const int sft = 1 << s_;
                    for (int t = lane; t < nx; t += 64) {
                        const int a = ((t >> s_) << (s_ + 1)) | sft;
                        const int ti = a + (t & (sft - 1)), si = a - 1;
                        sbuf[ti] = round_to<T>(sbuf[ti] + sbuf[si]);
                    }
                    wave_sync();
                }
                for (int e = lane; e < W; e += 64)
                    if (c0 + e < nk_img && sbuf[e] <= p_thr) ++count;
                total = sbuf[W - 1];
                wave_sync();
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) count += __shfl_xor(count, o);
    }
    int n = count + 1;
    if (n < top_k) n = top_k;
    if (n > nk_img) n = nk_img;

    wave_sync();        // bits zeroed above
    for (int i = lane; i < n; i += 64) {
        const int col = 0xFFFF - (int)(keys[i] & 0xFFFFu);
        atomicOr(&bits[col >> 5], 1u << (col & 31));
    }
    if (neighbors && m < nb_rows) {
        const int lim = nk_img < nb_cols ? nk_img : nb_cols;
        const uint8_t* nr = neighbors + (long long)m * nb_cols;
        for (int j = lane; j < lim; j += 64)
            if (nr[j]) atomicOr(&bits[j >> 5], 1u << (j & 31));
    }
    if (m < first_frame_blocks) {
        const int lim = first_frame_blocks < nk_all ? first_frame_blocks : nk_all;
        for (int j = lane; j < lim; j += 64) atomicOr(&bits[j >> 5], 1u << (j & 31));
    }
    for (int j = nk_img + lane; j < nk_all; j += 64) atomicOr(&bits[j >> 5], 1u << (j & 31));
    wave_sync();
    // ---- ascending compaction: exclusive popcount prefix over the (<= 80) words, two shuffle scans ----
    const int nwords = (nk_all + 31) >> 5;     // <= 80 (2048 image + 512 text columns)
    {
        const int c0 = lane < nwords ? __popc(bits[lane]) : 0;
        const int c1 = (lane < 16 && 64 + lane < nwords) ? __popc(bits[64 + lane]) : 0;
        int i0 = c0, i1 = c1;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t0 = __shfl_up(i0, o), t1 = __shfl_up(i1, o);
            if (lane >= o) {
                i0 += t0;
                i1 += t1;
            }
        }
        const int tot0 = __shfl(i0, 63);
        wpre[lane] = (uint32_t)(i0 - c0);
        if (lane < 16) wpre[64 + lane] = (uint32_t)(tot0 + i1 - c1);
        if (lane == 63 && cnt) cnt[row] = tot0 + i1;
    }
    wave_sync();
    for (int j = lane; j < nk_all && !(SEL_X & 8); j += 64) {
        const uint32_t w = bits[j >> 5];
        const bool on = (w >> (j & 31)) & 1u;
        if (mask) mask[row * nk_all + j] = on ? 1 : 0;
        if (on && idx) idx[row * nk_all + (int)wpre[j >> 5] + __popc(w & ((1u << (j & 31)) - 1u))] = j;
    }
    }   // rows of this wave
}


// ---- kept-count-aware launch order (SURVEY.md §7 "load imbalance -> work queue sorted by kept count") ------------
// order[bh][s * seg + j] = the query block of segment s (blocks [s * seg, (s+1) * seg)) with the j-th LARGEST kept
// count (ties: lower block first).  The attention kernel maps launch position -> query block through it, so that
// inside every XCD's contiguous range the long lists start first and the short ones fill the tail.
// Rank by counting: n <= 2048 per segment, one workgroup per (bh, segment).
__global__ void __launch_bounds__(256) order_by_count_kernel(const int32_t* __restrict__ cnt, int32_t* __restrict__ order,
                                                             int nq, int seg) {
    __shared__ int c_sh[2048];
    const int nseg = (nq + seg - 1) / seg;
    const long long bh = blockIdx.x / nseg;
    const int s0 = (int)(blockIdx.x % nseg) * seg;
    const int n = (s0 + seg <= nq) ? seg : nq - s0;
    for (int i = threadIdx.x; i < n; i += 256) c_sh[i] = cnt[bh * nq + s0 + i];
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) {
        const int ci = c_sh[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const int cj = c_sh[j];
            rank += (cj > ci) || (cj == ci && j < i);
        }
        order[bh * nq + s0 + rank] = s0 + i;
    }
}

}  // namespace
}  // namespace jenga

using namespace jenga;

extern "C" int jenga_block_select(void* stream, const void* qpool, const void* kpool, const uint8_t* neighbors,
                                  int64_t nb_rows, int64_t nb_cols, uint8_t* mask, int32_t* idx, int32_t* cnt,
                                  int64_t B, int64_t H, int64_t nq, int64_t nk_img, int64_t text_blocks, int64_t top_k,
                                  float p, int64_t first_frame_blocks, int dtype, int flags) {
    if (!qpool || !kpool || B < 0 || H < 0 || nq < 0 || nk_img <= 0 || text_blocks < 0 || top_k < 0) {
        set_error("jenga_block_select: bad arguments");
        return JENGA_EINVAL;
    }
    if (nk_img > 256 * MAX_PER_THREAD || nk_img + text_blocks > 80 * 32 || nk_img > 0xFFFF) {
        set_error("jenga_block_select: at most %d image key blocks supported (got %lld)", 256 * MAX_PER_THREAD,
                  (long long)nk_img);
        return JENGA_EUNSUPPORTED;
    }
    if (dtype != JENGA_BF16 && dtype != JENGA_FP16) {
        set_error("jenga_block_select: dtype must be bf16 or fp16");
        return JENGA_EUNSUPPORTED;
    }
    // float(head_dim ** -0.5) as torch multiplies a 16-bit tensor by the Python scalar (fp32 opmath)
    float scale;
    switch ((flags >> 8) & 0xFF) {
        case 0: case 128: scale = 0.08838834764831845f; break;
        case 64: scale = 0.125f; break;
        case 32: scale = 0.17677669529663687f; break;
        case 16: scale = 0.25f; break;
        default:
            set_error("jenga_block_select: JENGA_SELECT_HEAD_DIM must be 16, 32, 64 or 128 (got %d)", (flags >> 8) & 0xFF);
            return JENGA_EUNSUPPORTED;
    }
    const long long rows = (long long)B * H * nq;
    if (rows == 0) return JENGA_OK;
    int npow2 = 64;                  // the row, padded to 64 lanes x E keys (E a power of two)
    while (npow2 < nk_img) npow2 <<= 1;
    // device-scan mode: the chunk width torch's launcher would pick for a [rows, nk_img] tensor
    // (get_log_num_threads_x_inner_scan, ATen/native/cuda/ScanUtils.cuh:20-41)
    int scan_log_nx = -1;
    if (flags & JENGA_SELECT_DEVICE_SCAN) {
        uint32_t lx = 0, ly = 0;      // (torch instantiates the helper with uint32_t: the wrap-around is part of it)
        while ((1ULL << lx) < (unsigned long long)nk_img) ++lx;
        while ((1ULL << ly) < (unsigned long long)rows) ++ly;
        uint32_t l = ((uint32_t)9 + (lx - ly)) / (uint32_t)2;
        if (l < 4u) l = 4u;
        if (l > 9u) l = 9u;
        scan_log_nx = (int)l;
    }
    const size_t smem = ((size_t)SEL_R * npow2 > 8192 ? (size_t)SEL_R * npow2 : 8192) * 4 + (size_t)SEL_R * 80 * 4 * 2 +
                        (scan_log_nx > 5 ? (size_t)SEL_R * ((size_t)2 << scan_log_nx) * 4 : 0);
    // the reference compares the dtype cumsum with a Python float: the scalar is rounded to the tensor dtype
    float p_thr;
    if (dtype == JENGA_BF16) {
        uint32_t u;
        std::memcpy(&u, &p, 4);
        u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
        std::memcpy(&p_thr, &u, 4);
    } else {
        p_thr = (float)(_Float16)p;
    }
    const long long BH = (long long)B * H;
    const long long ngrp = (nq + SEL_R - 1) / SEL_R;
    const int xcd_map = (BH % 8 == 0) ? 1 : 0;
    const long long grid = BH * ngrp;            // (xcd_map: BH % 8 == 0, the same count, another order)
    if (grid > 0x7fffffffLL || BH > 0x7fffffffLL) {
        set_error("jenga_block_select: grid size %lld out of range", grid);
        return JENGA_EINVAL;
    }
    // the largest request (2048-column rows, device-scan mode included) is SEL_MAX_SMEM = 51 712 bytes with SEL_R = 4: under the
    // 64 KiB every launch gets without hipFuncSetAttribute, so no attribute call is needed
    constexpr int SEL_MAX_SMEM = SEL_R * 2048 * 4 + SEL_R * 80 * 4 * 2 + SEL_R * 1024 * 4;
    static_assert(SEL_MAX_SMEM <= 65536, "block_select_kernel would need hipFuncAttributeMaxDynamicSharedMemorySize");
#define LAUNCH_SEL(T, E_)                                                                                             \
    do {                                                                                                              \
        hipLaunchKernelGGL((block_select_kernel<T, E_>), dim3((unsigned)grid), dim3(256), smem, (hipStream_t)stream,  \
                           (const uint16_t*)qpool, (const uint16_t*)kpool, neighbors, (int)nb_rows, (int)nb_cols,     \
                           mask, idx, cnt, (int)BH, (int)nq, (int)nk_img, (int)text_blocks, (int)top_k, p_thr,        \
                           (int)first_frame_blocks, scan_log_nx, (int)ngrp, xcd_map, scale);                          \
    } while (0)
#define LAUNCH_SEL_E(E_)                                                                                              \
    do {                                                                                                              \
        if (dtype == JENGA_BF16) LAUNCH_SEL(BF16, E_); else LAUNCH_SEL(FP16, E_);                                     \
    } while (0)
    switch (npow2 >> 6) {
        case 1: LAUNCH_SEL_E(1); break;
        case 2: LAUNCH_SEL_E(2); break;
        case 4: LAUNCH_SEL_E(4); break;
        case 8: LAUNCH_SEL_E(8); break;
        case 16: LAUNCH_SEL_E(16); break;
        default: LAUNCH_SEL_E(32); break;
    }
#undef LAUNCH_SEL_E
#undef LAUNCH_SEL
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("jenga_block_select: %s", hipGetErrorString(e));
        return JENGA_ELAUNCH;
    }
    return JENGA_OK;
}

extern "C" int jenga_order_by_count(void* stream, const int32_t* cnt, int64_t BH, int64_t nq, int64_t segment,
                                    int32_t* order) {
    if (!cnt || !order || BH < 0 || nq < 0 || segment <= 0) {
        set_error("jenga_order_by_count: bad arguments");
        return JENGA_EINVAL;
    }
    if (segment > nq) segment = nq;
    if (segment > 2048) {
        set_error("jenga_order_by_count: at most 2048 query blocks per segment (got %lld)", (long long)segment);
        return JENGA_EUNSUPPORTED;
    }
    if (BH * nq == 0) return JENGA_OK;
    const long long nseg = (nq + segment - 1) / segment;
    hipLaunchKernelGGL(order_by_count_kernel, dim3((unsigned)(BH * nseg)), dim3(256), 0, (hipStream_t)stream, cnt, order,
                       (int)nq, (int)segment);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("jenga_order_by_count: %s", hipGetErrorString(e));
        return JENGA_ELAUNCH;
    }
    return JENGA_OK;
}
