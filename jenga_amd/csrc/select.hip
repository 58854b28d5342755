// Block selection: one workgroup per (batch, head, query block) row.
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

template <typename T>
__global__ void __launch_bounds__(256)
block_select_kernel(const uint16_t* __restrict__ qpool, const uint16_t* __restrict__ kpool,
                    const uint8_t* __restrict__ neighbors, int nb_rows, int nb_cols, uint8_t* __restrict__ mask,
                    int32_t* __restrict__ idx, int32_t* __restrict__ cnt, int /*H*/, int nq, int nk_img, int text_blocks,
                    int top_k, float p_thr, int first_frame_blocks, int npow2, int scan_log_nx) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* keys = reinterpret_cast<uint32_t*>(smem);                 // [npow2]
    float* qrow = reinterpret_cast<float*>(keys + npow2);               // [128]
    uint32_t* bits = reinterpret_cast<uint32_t*>(qrow + 128);           // [80]: up to 2048+ columns
    uint32_t* wpre = bits + 80;                                         // [80] exclusive popcount prefix
    float* scratch = reinterpret_cast<float*>(wpre + 80);               // [8]
    int* n_sh = reinterpret_cast<int*>(scratch + 8);                    // [1]
    float* sbuf = reinterpret_cast<float*>(n_sh + 4);                   // [2 << scan_log_nx] (device-scan mode, W > 64)

    const int nk_all = nk_img + text_blocks;
    const long long row = blockIdx.x;  // (b*H + h)*nq + m
    const int m = (int)(row % nq);
    const long long bh = row / nq;
    const int tid = threadIdx.x;

    if (tid < 128) qrow[tid] = to_f32<T>(qpool[row * 128 + tid]);
    if (tid < 80) bits[tid] = 0u;
    __syncthreads();

    // ---- pooled scores for the image columns (K3) ----
    const float scale = 0.08838834764831845f;  // float(128 ** -0.5)
    float sc[MAX_PER_THREAD];
    float lmax = -INFINITY;
#pragma unroll
    for (int i = 0; i < MAX_PER_THREAD; ++i) {
        const int j = tid + i * 256;
        sc[i] = -INFINITY;
        if (j < nk_img) {
            const uint4* kr = reinterpret_cast<const uint4*>(kpool + (bh * nk_all + j) * 128);
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                float f[8];
                unpack8<T>(kr[c], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc = fmaf(qrow[c * 8 + e], f[e], acc);
            }
            sc[i] = round_to<T>(round_to<T>(acc) * scale);
            lmax = fmaxf(lmax, sc[i]);
        }
    }
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
    if (npow2 == 1024) bitonic_sort_desc_regs<4>(keys, npow2, tid);
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
    if (scan_log_nx < 0) {
        // default contract = torch.cumsum of a 16-bit tensor on the CPU (what the reference's goldens were generated
        // with): sequential fp32 accumulation, each partial rounded to dtype
        if (tid == 0) {
            float acc = 0.f;
            int count = 0;
            for (int i = 0; i < nk_img; ++i) {
                acc = acc + to_f32<T>((uint16_t)(keys[i] >> 16));
                if (round_to<T>(acc) <= p_thr)
                    ++count;
                else
                    break;  // partial sums are non-decreasing
            }
            *n_sh = count;
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
    const int nwords = (nk_all + 31) >> 5;
    if (tid < nwords) {
        int pre = 0;
        for (int w = 0; w < tid; ++w) pre += __popc(bits[w]);
        wpre[tid] = (uint32_t)pre;
        if (tid == nwords - 1 && cnt) cnt[row] = pre + __popc(bits[tid]);
    }
    __syncthreads();
    for (int j = tid; j < nk_all; j += 256) {
        const uint32_t w = bits[j >> 5];
        const bool on = (w >> (j & 31)) & 1u;
        if (mask) mask[row * nk_all + j] = on ? 1 : 0;
        if (on && idx) idx[row * nk_all + (int)wpre[j >> 5] + __popc(w & ((1u << (j & 31)) - 1u))] = j;
    }
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
    const size_t smem = (size_t)npow2 * 4 + 128 * 4 + 80 * 4 * 2 + 8 * 4 + 16 +
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
#define LAUNCH_SEL(T)                                                                                                 \
    hipLaunchKernelGGL(block_select_kernel<T>, dim3((unsigned)rows), dim3(256), smem, (hipStream_t)stream,            \
                       (const uint16_t*)qpool, (const uint16_t*)kpool, neighbors, (int)nb_rows, (int)nb_cols, mask,   \
                       idx, cnt, (int)H, (int)nq, (int)nk_img, (int)text_blocks, (int)top_k, p_thr,                   \
                       (int)first_frame_blocks, npow2, scan_log_nx)
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
