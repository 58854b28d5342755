// Micro-benchmark (round 4): what a plain streaming kernel reaches on this box -- the practical roof next to the nominal
// 8 TB/s that roofline_secondary prices the row kernels against (gather, LayerNorm+modulate, norm/RoPE/pool, pack_v move
// 0.7-3 GB per launch at 4.9-5.7 TB/s).  Buffers of 708 MB (= [115200, 3072] bf16, the hidden state of the 720p video):
//   copy   out[i] = in[i]            16 B per lane per access, grid-stride; bytes = read + write
//   read   sum of in[i]              bytes = read
//   write  out[i] = c                bytes = write
// for several grid sizes and with / without nontemporal access.   hipcc --offload-arch=gfx950 -O3 -o hbm_copy hbm_copy.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <int MODE, bool NT>
__global__ void __launch_bounds__(256) k(const u4* __restrict__ in, u4* __restrict__ out, size_t n, unsigned* sink) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    u4 acc = {0, 0, 0, 0};
    for (; i < n; i += stride) {
        if (MODE == 0) {
            const u4 v = NT ? __builtin_nontemporal_load(in + i) : in[i];
            if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v;
        } else if (MODE == 1) {
            const u4 v = NT ? __builtin_nontemporal_load(in + i) : in[i];
            acc ^= v;
        } else {
            const u4 v = {(unsigned)i, 1u, 2u, 3u};
            if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v;
        }
    }
    if (MODE == 1 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) *sink = 1;
}

template <int MODE, bool NT>
static void run(const char* name, const u4* in, u4* out, size_t n, int grid, unsigned* sink) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<MODE, NT>), dim3(grid), dim3(256), 0, 0, in, out, n, sink);
    hipEventRecord(e0);
    const int reps = 20;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<MODE, NT>), dim3(grid), dim3(256), 0, 0, in, out, n, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double bytes = (MODE == 0 ? 2.0 : 1.0) * n * 16;
    printf("{\"kernel\": \"%s\", \"nontemporal\": %s, \"grid\": %d, \"ms\": %.4f, \"GBps\": %.1f, \"frac_of_8TBps\": %.3f}\n", name,
           NT ? "true" : "false", grid, ms, bytes / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 8e12);
}

int main() {
    const size_t bytes = (size_t)115200 * 3072 * 2, n = bytes / 16;
    u4 *in, *out;
    unsigned* sink;
    hipMalloc(&in, bytes);
    hipMalloc(&out, bytes);
    hipMalloc(&sink, 4);
    hipMemset(in, 1, bytes);
    hipMemset(out, 0, bytes);
    const int grids[] = {1024, 2048, 4096, 8192, 16384, (int)((n + 255) / 256)};
    for (int g : grids) {
        run<0, false>("copy", in, out, n, g, sink);
        run<0, true>("copy", in, out, n, g, sink);
    }
    for (int g : {2048, 8192}) {
        run<1, false>("read", in, out, n, g, sink);
        run<1, true>("read", in, out, n, g, sink);
        run<2, false>("write", in, out, n, g, sink);
        run<2, true>("write", in, out, n, g, sink);
    }
    // the runtime's own device-to-device copy
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, 0);
    hipEventRecord(e0);
    for (int r = 0; r < 20; ++r) hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 20;
    printf("{\"kernel\": \"hipMemcpyAsync D2D\", \"ms\": %.4f, \"GBps\": %.1f, \"frac_of_8TBps\": %.3f}\n", ms, 2.0 * bytes / (ms * 1e-3) / 1e9,
           2.0 * bytes / (ms * 1e-3) / 8e12);
    return 0;
}
