// Micro-benchmark: do MFMA and VALU work of DIFFERENT waves on the same SIMD overlap on gfx950?
// 8-wave workgroups (wave w and w+4 share a SIMD), one per CU.  mode 0: waves 0-3 MFMA, 4-7 idle;
// mode 1: waves 0-3 idle, 4-7 VALU; mode 2: waves 0-3 MFMA, 4-7 VALU; mode 3: all 8 waves alternate MFMA/VALU chunks.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ void __launch_bounds__(512, 2) k(float* out, int iters, int exp_per_iter) {
    const int wave = threadIdx.x >> 6;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    float v[32];
    for (int i = 0; i < 32; ++i) v[i] = threadIdx.x * 1e-3f + i;
    const bool do_mfma = (MODE == 0 || MODE == 2) ? (wave < 4) : (MODE == 3);
    const bool do_valu = (MODE == 1 || MODE == 2) ? (wave >= 4) : (MODE == 3);
    for (int it = 0; it < iters; ++it) {
        if (do_mfma) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        }
        if (do_valu) {
            for (int e = 0; e < exp_per_iter; ++e) {
#pragma unroll
                for (int i = 0; i < 32; ++i) v[i] = __builtin_amdgcn_exp2f(v[i] * 0.999f - 0.5f) + 1.0f;
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 32; ++i) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE> float run(float* d, int iters, int epi) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, iters, epi);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, iters, epi);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    const int iters = 20000;
    for (int epi = 1; epi <= 2; ++epi) {
        float m0 = run<0>(d, iters, epi), m1 = run<1>(d, iters, epi), m2 = run<2>(d, iters, epi), m3 = run<3>(d, iters, epi);
        // per iteration: 32 MFMAs (1024 pipe cycles) vs epi*32*(mul,exp,add)
        printf("valu_chunks=%d  mfma_only %.2f ms  valu_only %.2f ms  split_waves(both) %.2f ms  all_waves_alternate %.2f ms  | sum %.2f max %.2f\n",
               epi, m0, m1, m2, m3, m0 + m1, m0 > m1 ? m0 : m1);
    }
    return 0;
}
