// Block selection: one workgroup per SEL_G consecutive query blocks of one (batch, head) (round 5; one row per workgroup
// before).  The pooled K of a head (902 x 128 at the 720p shape = 230 KB) is what every row of the head reads: with one row
// per workgroup that was 5 GB per launch through the L2s, and since workgroup ids go round robin over the 8 XCDs every XCD's
// 4 MB L2 saw all 24 heads (5.5 MB).  Now a workgroup streams its head's pooled K ONCE for SEL_G rows (the dot products keep
// their order: per (row, column) the same 128 sequential fmaf), and with (batch * heads) a multiple of 8 all workgroups of a
// head run on ONE XCD (id % 8), so the re-reads are L2 hits.  Measured (profiles/r05_select_ab.json): 0.70 -> 0.49 ms at the
// 720p shape; of the 0.70 the one-thread cumulative sum was 0.26 on flat rows (now an exact shuffle scan), the dot products
// ~0.15, the sort ~0.18, launch + output ~0.1; 2 rows per workgroup is the optimum (4: 0.53-0.73, 8: 0.58-0.77 -- the rows of
// a group run one after the other, more rows = fewer workgroups to hide the sort's barriers).  The per-row work:
//   scores -> softmax (dtype) -> sort (bitonic, LDS) -> cumulative-probability / top-k rule -> bit set
//   -> OR static neighbours / first-frame / text columns -> ascending index list (+ optional one-hot mask).
// Rounding points follow the reference's torch code running in the tensor dtype
// (attention_block_triton_diffres.py:221-250); see include/jenga_amd.h.
#include <cstring>

#include "common.h"

namespace jenga {
namespace {

__device__ __forceinline__ float block_reduce_max(float v, float* scratch) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = scratch[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = fmaxf(r, scratch[i]);
    __syncthreads();
    return r;
}
__device__ __forceinline__ float block_reduce_sum(float v, float* scratch) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = scratch[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r += scratch[i];
    __syncthreads();
    return r;
}

constexpr int MAX_PER_THREAD = 8;  // 256 threads x 8 = up to 2048 image key blocks

// Bitonic sort (descending) of npow2 = 256 * E unique 32-bit keys held in LDS, E consecutive keys per thread in
// registers: compare-exchange distances below E stay inside a thread, distances below 64 * E are wave shuffles, and
// only the two or three largest distances go through LDS with a barrier (the plain LDS form below pays 55 barriers
// at 1024 keys).  Keys are unique, so every correct sort gives the same order: results are unchanged bit for bit.
template <int E>
__device__ __forceinline__ void bitonic_sort_desc_regs(uint32_t* keys, int npow2, int tid) {
    uint32_t v[E];
#pragma unroll
    for (int r = 0; r < E; ++r) v[r] = keys[tid * E + r];
    for (int k = 2; k <= npow2; k <<= 1) {
        int j = k >> 1;
        for (; j >= E; j >>= 1) {   // partner element lives in thread tid ^ (j / E), same register index
            const int pj = j / E;
            uint32_t pv[E];
            if (pj < 64) {
#pragma unroll
                for (int r = 0; r < E; ++r) pv[r] = (uint32_t)__shfl_xor((int)v[r], pj);
            } else {
                __syncthreads();
#pragma unroll
                for (int r = 0; r < E; ++r) keys[tid * E + r] = v[r];
                __syncthreads();
#pragma unroll
                for (int r = 0; r < E; ++r) pv[r] = keys[(tid ^ pj) * E + r];
            }
            const int i0 = tid * E;   // (i & j) and (i & k) do not depend on r here: j, k >= E
            const bool take_max = (((i0 & j) == 0) == ((i0 & k) == 0));
#pragma unroll
            for (int r = 0; r < E; ++r) v[r] = take_max ? (v[r] > pv[r] ? v[r] : pv[r]) : (v[r] < pv[r] ? v[r] : pv[r]);
        }
#pragma unroll
        for (int jj = E / 2; jj > 0; jj >>= 1) {   // inside the thread, compile-time register indices
            if (jj < k) {
#pragma unroll
                for (int r = 0; r < E; ++r) {
                    if ((r & jj) == 0) {
                        const int i = tid * E + r;
                        const bool desc = ((i & k) == 0);
                        const uint32_t a = v[r], b = v[r | jj];
                        const bool sw = desc ? (a < b) : (a > b);
                        v[r] = sw ? b : a;
                        v[r | jj] = sw ? a : b;
                    }
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < E; ++r) keys[tid * E + r] = v[r];
    __syncthreads();
}

#ifndef SEL_G_ROWS
#define SEL_G_ROWS 2
#endif
constexpr int SEL_G = SEL_G_ROWS;  // query blocks per workgroup
// elimination switches (A/B builds, wrong results): SEL_X & 1 no dot products, 2 no sort, 4 no cumulative sum, 8 no output
#ifndef SEL_X
#define SEL_X 0
#endif

template <typename T>
__global__ void __launch_bounds__(256)
block_select_kernel(const uint16_t* __restrict__ qpool, const uint16_t* __restrict__ kpool,
                    const uint8_t* __restrict__ neighbors, int nb_rows, int nb_cols, uint8_t* __restrict__ mask,
                    int32_t* __restrict__ idx, int32_t* __restrict__ cnt, int BH, int nq, int nk_img, int text_blocks,
                    int top_k, float p_thr, int first_frame_blocks, int npow2, int scan_log_nx, int ngrp, int xcd_map) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* keys = reinterpret_cast<uint32_t*>(smem);                 // [npow2]
    float* qrows = reinterpret_cast<float*>(keys + npow2);              // [SEL_G][128]
    uint32_t* bits = reinterpret_cast<uint32_t*>(qrows + SEL_G * 128);  // [80]: up to 2048+ columns
    uint32_t* wpre = bits + 80;                                         // [80] exclusive popcount prefix
    float* scratch = reinterpret_cast<float*>(wpre + 80);               // [8]
    int* n_sh = reinterpret_cast<int*>(scratch + 8);                    // [4]
    float* sbuf = reinterpret_cast<float*>(n_sh + 4);                   // [2 << scan_log_nx] (device-scan mode, W > 64)

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
    const int m0 = grp * SEL_G;
    const int g_n = nq - m0 < SEL_G ? nq - m0 : SEL_G;      // rows of this group

    for (int e = tid; e < SEL_G * 128; e += 256) {
        const int g = e >> 7;
        qrows[e] = g < g_n ? to_f32<T>(qpool[(bh * nq + m0 + g) * 128 + (e & 127)]) : 0.f;
    }
    __syncthreads();

    // ---- pooled scores for the image columns (K3): the K row of a column is read once for the SEL_G rows ----
    const float scale = 0.08838834764831845f;  // float(128 ** -0.5)
    float scg[SEL_G][MAX_PER_THREAD];
#pragma unroll
    for (int i = 0; i < MAX_PER_THREAD; ++i) {
        const int j = tid + i * 256;
#pragma unroll
        for (int g = 0; g < SEL_G; ++g) scg[g][i] = -INFINITY;
        if (j < nk_img && (SEL_X & 1)) {
#pragma unroll
            for (int g = 0; g < SEL_G; ++g) scg[g][i] = (float)((j * 37 + g * 11) % 97) * 0.01f;
        } else if (j < nk_img) {
            const uint4* kr = reinterpret_cast<const uint4*>(kpool + (bh * nk_all + j) * 128);
            float acc[SEL_G];
#pragma unroll
            for (int g = 0; g < SEL_G; ++g) acc[g] = 0.f;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                float f[8];
                unpack8<T>(kr[c], f);
#pragma unroll
                for (int g = 0; g < SEL_G; ++g)
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[g] = fmaf(qrows[g * 128 + c * 8 + e], f[e], acc[g]);
            }
#pragma unroll
            for (int g = 0; g < SEL_G; ++g) scg[g][i] = round_to<T>(round_to<T>(acc[g]) * scale);
        }
    }
#pragma unroll 1
    for (int g = 0; g < g_n; ++g) {
    const int m = m0 + g;
    const long long row = bh * nq + m;
    if (tid < 80) bits[tid] = 0u;
    float sc[MAX_PER_THREAD];
    float lmax = -INFINITY;
#pragma unroll
    for (int i = 0; i < MAX_PER_THREAD; ++i) {
        sc[i] = scg[0][i];
#pragma unroll
        for (int g2 = 1; g2 < SEL_G; ++g2) sc[i] = g == g2 ? scg[g2][i] : sc[i];
        lmax = fmaxf(lmax, sc[i]);
    }
    __syncthreads();
    // ---- softmax over the image columns, rounded to dtype (K4) ----
    const float rmax = block_reduce_max(lmax, scratch);
    float lsum = 0.f;
#pragma unroll
    for (int i = 0; i < MAX_PER_THREAD; ++i) {
        const int j = tid + i * 256;
        if (j < nk_img) {
            sc[i] = expf(sc[i] - rmax);
            lsum += sc[i];
        }
    }
    const float rsum = block_reduce_sum(lsum, scratch);
#pragma unroll
    for (int i = 0; i < MAX_PER_THREAD; ++i) {
        const int j = tid + i * 256;
        if (j < npow2) {
            uint32_t key = 0u;  // padding sorts last
            if (j < nk_img) key = ((uint32_t)from_f32<T>(sc[i] / rsum) << 16) | (uint32_t)(0xFFFF - j);
            keys[j] = key;
        }
    }
    __syncthreads();
    // ---- bitonic sort, descending on (probability bits, then lower column first) ----
    if (SEL_X & 2) {
    } else if (npow2 == 1024) bitonic_sort_desc_regs<4>(keys, npow2, tid);
    else if (npow2 == 512) bitonic_sort_desc_regs<2>(keys, npow2, tid);
    else if (npow2 == 256) bitonic_sort_desc_regs<1>(keys, npow2, tid);
    else if (npow2 == 2048) bitonic_sort_desc_regs<8>(keys, npow2, tid);
    else
    for (int k = 2; k <= npow2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < npow2; i += 256) {
                const int l = i ^ j;
                if (l > i) {
                    const uint32_t a = keys[i], b = keys[l];
                    const bool desc = ((i & k) == 0);
                    if (desc ? (a < b) : (a > b)) {
                        keys[i] = b;
                        keys[l] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
    // ---- n = max(#(cumsum <= p) + 1, top_k) ----
    if (SEL_X & 4) {
        if (tid == 0) *n_sh = top_k;
    } else if (scan_log_nx < 0) {
        // default contract = torch.cumsum of a 16-bit tensor on the CPU (what the reference's goldens were generated
        // with): sequential fp32 accumulation, each partial rounded to dtype
        // Round 5.  Wave 0 takes 64 sorted probabilities at a time (one LDS read per lane).
        //  * Fast path, bit-identical by construction: the probabilities are 16-bit values and the partial sums stay below
        //    2, so as long as every addend so far is at least 2^-16 (bf16; 2^-13 for fp16) every partial sum -- in ANY order
        //    of addition -- is a multiple of 2^-24 below 2 and therefore exact in fp32: a shuffle scan gives the sequential
        //    chain's values.  The list is sorted descending, so one look at the chunk's last element decides.
        //  * Otherwise (tiny probabilities before the threshold is reached: p near 1, very peaky rows) the chunk runs the
        //    sequential chain itself, every lane redundantly on values broadcast with v_readlane.
        // The partial sums are non-decreasing, so "count while round(acc) <= p, stop at the first miss" equals "count every
        // i with round(acc_i) <= p"; a chunk whose last partial already missed ends the walk.
        if (tid < 64) {
            const float exact_min = sizeof(T) && __is_same(T, BF16) ? 1.52587890625e-05f : 1.220703125e-04f;
            float acc = 0.f;
            int count = 0;
            for (int c0 = 0; c0 < nk_img; c0 += 64) {
                const int col = c0 + tid;
                const uint32_t mine = col < nk_img ? (keys[col] >> 16) : 0u;
                const float pv = to_f32<T>((uint16_t)mine);
                const int last_col = c0 + 63 < nk_img ? 63 : nk_img - 1 - c0;
                const float smallest = __shfl(pv, last_col);
                if (smallest >= exact_min) {
                    float inc = pv;
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) {
                        const float t_ = __shfl_up(inc, o);
                        if (tid >= o) inc += t_;
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
            if (tid == 0) *n_sh = count;
        }
    } else {
        // JENGA_SELECT_DEVICE_SCAN: torch.cumsum of a 16-bit tensor on the DEVICE, restated (ATen/native/cuda/
        // ScanUtils.cuh, tensor_kernel_scan_innermost_dim_impl, torch 2.10): the row is scanned in chunks of
        // W = 2 * nx columns (nx = 2^scan_log_nx threads, chosen by get_log_num_threads_x_inner_scan from the number
        // of rows and the row length); inside a chunk a Sklansky network, EVERY add rounded to the 16-bit dtype
        // (row_buf is scalar_t); the chunk's last element (also 16-bit) is added to the next chunk's first one.
        // Partials are not monotonic, so ALL columns with cumsum <= p are counted ((cumsum <= p).sum(), :244-245).
        const int nx = 1 << scan_log_nx, W = 2 * nx;
        if (tid == 0) *n_sh = 0;
        __syncthreads();
        if (W <= 64) {
            if (tid < 64) {     // one wave, the chunk in registers, Sklansky steps as wave shuffles
                float total = 0.f;
                int count = 0;
                for (int c0 = 0; c0 < nk_img; c0 += W) {
                    const int col = c0 + tid;
                    float v = (tid < W && col < nk_img) ? to_f32<T>((uint16_t)(keys[col] >> 16)) : 0.f;
                    if (tid == 0) v = round_to<T>(v + total);
                    for (int m = 0; m <= scan_log_nx; ++m) {
                        const int sft = 1 << m;
                        const int src = (tid & ~(2 * sft - 1)) + sft - 1;
                        const float o = __shfl(v, src & 63);
                        if (tid & sft) v = round_to<T>(v + o);
                    }
                    if (tid < W && col < nk_img && v <= p_thr) ++count;
                    total = __shfl(v, W - 1);
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) count += __shfl_xor(count, o);
                if (tid == 0) *n_sh = count;
            }
        } else {
            float total = 0.f;
            int count = 0;
            for (int c0 = 0; c0 < nk_img; c0 += W) {
                for (int e = tid; e < W; e += 256)
                    sbuf[e] = (c0 + e < nk_img) ? to_f32<T>((uint16_t)(keys[c0 + e] >> 16)) : 0.f;
                __syncthreads();
                if (tid == 0) sbuf[0] = round_to<T>(sbuf[0] + total);
                __syncthreads();
                for (int m = 0; m <= scan_log_nx; ++m) {
                    const int sft = 1 << m;
                    for (int t = tid; t < nx; t += 256) {
                        const int a = ((t >> m) << (m + 1)) | sft;
                        const int ti = a + (t & (sft - 1)), si = a - 1;
                        sbuf[ti] = round_to<T>(sbuf[ti] + sbuf[si]);
                    }
                    __syncthreads();
                }
                for (int e = tid; e < W; e += 256)
                    if (c0 + e < nk_img && sbuf[e] <= p_thr) ++count;
                total = sbuf[W - 1];
                __syncthreads();
            }
            if (count) atomicAdd(n_sh, count);
        }
    }
    __syncthreads();
    if (tid == 0) {
        int n = *n_sh + 1;
        if (n < top_k) n = top_k;
        if (n > nk_img) n = nk_img;
        *n_sh = n;
    }
    __syncthreads();
    const int n = *n_sh;
    for (int i = tid; i < n; i += 256) {
        const int col = 0xFFFF - (int)(keys[i] & 0xFFFFu);
        atomicOr(&bits[col >> 5], 1u << (col & 31));
    }
    if (neighbors && m < nb_rows) {
        const int lim = nk_img < nb_cols ? nk_img : nb_cols;
        const uint8_t* nr = neighbors + (long long)m * nb_cols;
        for (int j = tid; j < lim; j += 256)
            if (nr[j]) atomicOr(&bits[j >> 5], 1u << (j & 31));
    }
    if (m < first_frame_blocks) {
        const int lim = first_frame_blocks < nk_all ? first_frame_blocks : nk_all;
        for (int j = tid; j < lim; j += 256) atomicOr(&bits[j >> 5], 1u << (j & 31));
    }
    for (int j = nk_img + tid; j < nk_all; j += 256) atomicOr(&bits[j >> 5], 1u << (j & 31));
    __syncthreads();
    // ---- ascending compaction ----
    const int nwords = (nk_all + 31) >> 5;     // <= 80 (2048 image + 512 text columns)
    if (tid < 128) {          // exclusive popcount prefix over the words: two waves, a shuffle scan each, the carry through LDS
        const int c = tid < nwords ? __popc(bits[tid]) : 0;
        int inc = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t_ = __shfl_up(inc, o);
            if ((tid & 63) >= o) inc += t_;
        }
        if (tid < 80) wpre[tid] = (uint32_t)(inc - c);
        if (tid == 63) n_sh[1] = inc;         // popcount of words 0..63
    }
    __syncthreads();
    if (tid >= 64 && tid < 80) wpre[tid] += (uint32_t)n_sh[1];
    __syncthreads();
    if (tid == nwords - 1 && cnt) cnt[row] = (int)wpre[tid] + __popc(bits[tid]);
    for (int j = tid; j < nk_all && !(SEL_X & 8); j += 256) {
        const uint32_t w = bits[j >> 5];
        const bool on = (w >> (j & 31)) & 1u;
        if (mask) mask[row * nk_all + j] = on ? 1 : 0;
        if (on && idx) idx[row * nk_all + (int)wpre[j >> 5] + __popc(w & ((1u << (j & 31)) - 1u))] = j;
    }
    __syncthreads();     // keys / bits / wpre / n_sh are reused by the next row
    }   // rows of the group
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
    const long long rows = (long long)B * H * nq;
    if (rows == 0) return JENGA_OK;
    int npow2 = 2;
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
    const size_t smem = (size_t)npow2 * 4 + SEL_G * 128 * 4 + 80 * 4 * 2 + 8 * 4 + 16 +
                       (scan_log_nx > 5 ? ((size_t)2 << scan_log_nx) * 4 : 0);
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
    const long long ngrp = (nq + SEL_G - 1) / SEL_G;
    const int xcd_map = (BH % 8 == 0) ? 1 : 0;
    const long long grid = BH * ngrp;            // (xcd_map: BH % 8 == 0, the same count, another order)
    if (grid > 0x7fffffffLL || BH > 0x7fffffffLL) {
        set_error("jenga_block_select: grid size %lld out of range", grid);
        return JENGA_EINVAL;
    }
#define LAUNCH_SEL(T)                                                                                                 \
    hipLaunchKernelGGL(block_select_kernel<T>, dim3((unsigned)grid), dim3(256), smem, (hipStream_t)stream,            \
                       (const uint16_t*)qpool, (const uint16_t*)kpool, neighbors, (int)nb_rows, (int)nb_cols, mask,   \
                       idx, cnt, (int)BH, (int)nq, (int)nk_img, (int)text_blocks, (int)top_k, p_thr,                  \
                       (int)first_frame_blocks, npow2, scan_log_nx, (int)ngrp, xcd_map)
    if (dtype == JENGA_BF16) LAUNCH_SEL(BF16); else LAUNCH_SEL(FP16);
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
