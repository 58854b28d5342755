// Shared device helpers for the gfx950 kernels (wave = 64 lanes, hard-coded).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

#include "../../include/jenga_amd.h"

namespace jenga {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct BF16 {};  // tag types: storage is always 16-bit
struct FP16 {};

// ---- scalar conversions (round-to-nearest-even, identical to torch's .to(dtype)) ----
template <typename T> __device__ __forceinline__ float to_f32(uint16_t u);
template <> __device__ __forceinline__ float to_f32<BF16>(uint16_t u) { return __uint_as_float((uint32_t)u << 16); }
template <> __device__ __forceinline__ float to_f32<FP16>(uint16_t u) {
    return (float)__builtin_bit_cast(_Float16, u);
}
template <typename T> __device__ __forceinline__ uint16_t from_f32(float f);
template <> __device__ __forceinline__ uint16_t from_f32<BF16>(float f) {
    return __builtin_bit_cast(uint16_t, (__bf16)f);
}
template <> __device__ __forceinline__ uint16_t from_f32<FP16>(float f) {
    return __builtin_bit_cast(uint16_t, (_Float16)f);
}
// two values -> one packed dword (lo in bits 0..15).  Vector conversions so that bf16 becomes ONE
// v_cvt_pk_bf16_f32 (round-to-nearest-even) instead of two scalar converts + shift + or.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ uint32_t pack2(float lo, float hi);
template <> __device__ __forceinline__ uint32_t pack2<BF16>(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
template <> __device__ __forceinline__ uint32_t pack2<FP16>(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
}
template <typename T> __device__ __forceinline__ float round_to(float f) { return to_f32<T>(from_f32<T>(f)); }

template <typename T> __device__ __forceinline__ void unpack8(const uint4& v, float (&f)[8]) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = to_f32<T>((uint16_t)(w[i] & 0xffffu));
        f[2 * i + 1] = to_f32<T>((uint16_t)(w[i] >> 16));
    }
}
template <typename T> __device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
    uint4 v;
    v.x = pack2<T>(f[0], f[1]);
    v.y = pack2<T>(f[2], f[3]);
    v.z = pack2<T>(f[4], f[5]);
    v.w = pack2<T>(f[6], f[7]);
    return v;
}

// ---- MFMA 32x32x16, fp32 accumulate; operands are 8 x 16-bit per lane held in a uint4 ----
// C/D layout (guide §3): col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
template <typename T> __device__ __forceinline__ f32x16 mfma32(const uint4& a, const uint4& b, const f32x16& c);
template <> __device__ __forceinline__ f32x16 mfma32<BF16>(const uint4& a, const uint4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0,
                                                   0, 0);
}
template <> __device__ __forceinline__ f32x16 mfma32<FP16>(const uint4& a, const uint4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0,
                                                  0);
}

// Key order inside a 32-key group as the P.V product consumes it: position p = hi*16 + s*8 + j  <->  key
// crow(8*s + j, hi) with crow(r, hi) = (r&3) + 8*(r>>2) + 4*hi  (the row a lane holds in MFMA C-register r).
__host__ __device__ __forceinline__ int pv_key_of_pos(int p) {
    const int hi = (p >> 4) & 1, r = p & 15;
    return (r & 3) + 8 * (r >> 2) + 4 * hi;
}

void set_error(const char* fmt, ...);

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per kernel instantiation and device: `done` is a static of the
// instantiation's launch function
inline void lp_set_smem_once(const void* kernel, int bytes, bool (&done)[64]) {
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
    std::lock_guard<std::mutex> lock(mu);
    if (dev >= 0 && done[dev]) return;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess && dev >= 0)
        done[dev] = true;
}

}  // namespace jenga

// bsattn3.hip: the launcher behind jenga_bsattn_fwd(..., flags & JENGA_ATTN_LP); arguments already validated
int jenga_bsattn_lp_launch(void* stream, const void* q, const void* k, const void* vt, void* o, const int32_t* seqlens,
                           const int32_t* idx, const int32_t* cnt, const int32_t* order, int64_t B, int64_t H,
                           int64_t n_blocks, int64_t nq_img, int64_t q_sb, int64_t q_ss, int64_t q_sh, int64_t k_sb,
                           int64_t k_ss, int64_t k_sh, int64_t o_sb, int64_t o_ss, int64_t o_sh, float sm_scale,
                           float text_amp, int64_t text_block_start, int dtype, int flags);
