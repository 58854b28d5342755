// Micro-benchmark: how many bytes per clock per CU does the global -> LDS path deliver on gfx950, for the access
// pattern of the attention kernel (4-wave workgroups, 2 per CU, each wave issuing 8 x global_load_lds_dwordx4 = 8 KiB
// per 32 KiB tile, 3 tiles in flight), with NO compute?  Source: a buffer of `span` bytes per workgroup-group that is
// either L2-resident (small span) or streams from MALL/HBM (large span).  MODE 1 uses global_load_dwordx4 into
// registers + ds_write_b128 instead.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ void dma8(const void* base, unsigned lds, unsigned off) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %3, %2\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "global_load_lds_dwordx4 %3, %2 offset:1024\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "global_load_lds_dwordx4 %3, %2 offset:2048\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "global_load_lds_dwordx4 %3, %2 offset:3072\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds), "s"(base), "v"(off)
        : "memory", "scc");
}

template <int MODE>
__global__ void __launch_bounds__(256, 2) k(const unsigned char* src, size_t span_tiles, int tiles, float* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned smem_base =
        __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem);
    // each workgroup walks its own pseudo-random tile sequence inside [0, span_tiles)
    size_t t = (size_t)blockIdx.x * 7919u % span_tiles;
    float acc = 0.f;
    for (int i = 0; i < tiles; ++i) {
        const unsigned char* tile = src + t * 32768 + wave * 8192;
        const unsigned slot = smem_base + (i % 2) * 32768 + wave * 8192;
        if (MODE == 0 || MODE == 2 || MODE == 3) {
            // MODE 2: the attention kernel's source-side XOR swizzle (lane -> row lane>>4, chunk (lane&15)^row) inside
            // each contiguous 1 KiB; MODE 3: the same with rows 6 KiB apart ([S,H,D] K layout, 24 heads)
            unsigned off = lane * 16;
            if (MODE == 2) off = (lane >> 4) * 256 + (((lane & 15) ^ ((lane >> 4) + 1)) << 4);
            if (MODE == 3) off = (lane >> 4) * 6144 + (((lane & 15) ^ ((lane >> 4) + 1)) << 4);
            dma8(tile, slot, off);
            dma8(tile + 4096, slot + 4096, off);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // previous tile landed, this one in flight
        } else {
            uint4 r[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = *reinterpret_cast<const uint4*>(tile + j * 1024 + lane * 16);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                *reinterpret_cast<uint4*>(smem + (i % 2) * 32768 + wave * 8192 + j * 1024 + lane * 16) = r[j];
        }
        __syncthreads();
        acc += reinterpret_cast<const float*>(smem)[((i + 1) % 2) * 8192 + threadIdx.x];   // touch the landed tile
        t = (t * 1664525u + 1013904223u) % span_tiles;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE> void run(const unsigned char* src, size_t span_tiles, int tiles, float* out, const char* name) {
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(512), dim3(256), 65536, 0, src, span_tiles, tiles, out);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(512), dim3(256), 65536, 0, src, span_tiles, tiles, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 512.0 * tiles * 32768;
    printf("%-28s span %8.1f MB: %7.3f ms  %6.2f TB/s  = %5.1f B/clk/CU at 2.0 GHz\n", name, span_tiles * 32768 / 1e6, ms,
           bytes / ms / 1e9, bytes / 256 / (ms * 1e-3 * 2.0e9));
}
int main(int argc, char** argv) {
    const size_t maxbytes = (2048ull << 20) + (1 << 20);
    unsigned char* src; hipMalloc(&src, maxbytes); hipMemset(src, 1, maxbytes);
    float* out; hipMalloc(&out, 512 * 256 * 4);
    const int tiles = 2000;
    if (argc > 1) {   // "sweep": the delivery rate against the size of the window the tiles are drawn from
        for (size_t mb : {1ul, 2ul, 4ul, 8ul, 16ul, 24ul, 32ul, 48ul, 64ul, 96ul, 128ul, 192ul, 256ul, 384ul, 512ul, 2048ul})
            run<0>(src, (mb << 20) / 32768, tiles, out, "LDS-DMA");
        return 0;
    }
    for (size_t mb : {1ul, 16ul, 128ul, 2048ul}) {
        run<0>(src, (mb << 20) / 32768, tiles, out, "LDS-DMA");
        run<2>(src, (mb << 20) / 32768, tiles, out, "LDS-DMA, XOR-swizzled source");
        run<3>(src, (mb << 20) / 32768, tiles, out, "LDS-DMA, swizzled + 6 KiB rows");
        run<1>(src, (mb << 20) / 32768, tiles, out, "global_load + ds_write");
    }
    return 0;
}
