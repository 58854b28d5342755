// Load generators for tools/diag_streams.py: kernels that only issue MFMAs (registers, no memory traffic).
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/micro/mfma_spin.hip -o tools/micro/bin/libmfma_spin.so
#include <hip/hip_runtime.h>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ void __launch_bounds__(256) spin(float* sink, int iters) {
    bf8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x - i)); }
    if (KIND == 0) {
        f16v acc = {0};
        for (int it = 0; it < iters; ++it) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc, 0, 0, 0);
        }
        if (acc[0] == 12345.678f) sink[0] = acc[1];
    } else if (KIND == 1) {
        f4v acc = {0};
        for (int it = 0; it < iters; ++it) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, acc, 0, 0, 0);
        }
        if (acc[0] == 12345.678f) sink[0] = acc[1];
    } else {            // no MFMA: a scalar fma chain of the same length
        float acc = 0.f, x = 0.001f * threadIdx.x;
        for (int it = 0; it < iters * 16; ++it) acc = __builtin_fmaf(acc, x, 1.0f);
        if (acc == 12345.678f) sink[0] = acc;
    }
}
extern "C" void mfma_spin(void* stream, float* sink, int kind, int blocks, int iters) {
    if (kind == 0) hipLaunchKernelGGL(spin<0>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, sink, iters);
    else if (kind == 1) hipLaunchKernelGGL(spin<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, sink, iters);
    else hipLaunchKernelGGL(spin<2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, sink, iters);
}
