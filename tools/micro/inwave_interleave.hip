// Micro-benchmark: can ONE wave hide its own softmax-like VALU work under its own MFMAs when the two are interleaved
// in the instruction stream (gfx950)?  Per iteration: 32 MFMA 32x32x16 (4 independent chains, 1024 pipe cycles) and a
// softmax-like bundle on 32 values (32 v_exp, 32 int add, 32 int max, 32 f32 add, 16 cvt_pk).
//   MODE 0: MFMA only      MODE 1: VALU only      MODE 2: MFMA block, sched_barrier, VALU block (dependent phases, as in attention)
//   MODE 3: the same two blocks with no barrier: the scheduler interleaves them (independent work)
// WAVES_PER_SIMD = 1 or 2 (256- or 512-thread workgroups, one per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE, int THREADS>
__global__ void __launch_bounds__(THREADS, 1) k(float* out, int iters, int nshift_in) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    float v[32];
    for (int i = 0; i < 32; ++i) v[i] = threadIdx.x * 1e-3f + i * 0.01f;
    float psum = 0.f;
    unsigned pk = 0;
    const int nshift = nshift_in;
    for (int it = 0; it < iters; ++it) {
        if (MODE != 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        }
        if (MODE == 2) __builtin_amdgcn_sched_barrier(0);  // keep the two blocks apart
        if (MODE != 0) {
            float p[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                p[i] = __int_as_float(max(__float_as_int(__builtin_amdgcn_exp2f(v[i])) + nshift, 0));
                psum += p[i];
            }
#pragma unroll
            for (int i = 0; i < 32; i += 2) {
                f32x2 t = {p[i], p[i + 1]};
                bf16x2 h = __builtin_convertvector(t, bf16x2);
                pk ^= *reinterpret_cast<unsigned*>(&h);
            }
#pragma unroll
            for (int i = 0; i < 32; ++i) v[i] = __int_as_float(__float_as_int(v[i]) ^ (pk & 1));  // cheap dependence
        }
    }
    float s = psum + (float)pk;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 32; ++i) s += v[i];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
}

template <int MODE, int THREADS> float run(float* d, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, THREADS>), dim3(256), dim3(THREADS), 0, 0, d, iters, 1 << 23);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, THREADS>), dim3(256), dim3(THREADS), 0, 0, d, iters, 1 << 23);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    const int iters = 20000;
    printf("1 wave/SIMD : mfma %.2f  valu %.2f  sequential %.2f  interleaved %.2f ms\n", run<0, 256>(d, iters), run<1, 256>(d, iters),
           run<2, 256>(d, iters), run<3, 256>(d, iters));
    printf("2 waves/SIMD: mfma %.2f  valu %.2f  sequential %.2f  interleaved %.2f ms (twice the work)\n", run<0, 512>(d, iters),
           run<1, 512>(d, iters), run<2, 512>(d, iters), run<3, 512>(d, iters));
    return 0;
}
