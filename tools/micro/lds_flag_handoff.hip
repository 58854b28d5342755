// Micro-benchmark (round 4): what does it cost to hand a staged K/V tile from its producers to its
// consumers through per-slot READY / CONSUMED counters in LDS instead of a workgroup-wide s_barrier -- the handoff a
// "one 8-wave workgroup per CU, deep shared ring, two decoupled 4-wave consumer groups" attention kernel would need.
//
// One 512-thread workgroup per CU (160 KiB LDS): a ring of R = 8 slots of 16 KiB (one 64-key K tile), every wave stages
// its 2 KiB of tile t+6 by LDS-DMA (global_load_lds_dwordx4, the attention kernel's staging), every wave consumes the
// whole tile t (16 ds_read_b128 feeding `mfma` MFMAs 32x32x16 bf16).  Per tile a wave executes
//   MODE 0  s_waitcnt vmcnt(my piece of t) ; s_barrier                               (the LP kernel's handoff, 8 waves)
//   MODE 1  s_waitcnt vmcnt ; ds_add ready[t%R] ; poll ready[t%R] >= 8 * epoch ; ... ; ds_add consumed[t%R]
//           and, before overwriting a slot, poll consumed[slot] >= 8 * epoch         (flags, read when needed)
//   MODE 2  as 1, with the flag reads issued ONE STEP AHEAD (the value is in a register when it is checked)
// `jitter` makes the two groups (waves 0-3, 4-7) do a different, pseudo-random number of MFMAs per tile (the "blocks only
// one of the two query blocks keeps"): under a barrier a step costs max(A, B), with flags the slack of the ring absorbs it.
//   hipcc --offload-arch=gfx950 -O3 -o lds_flag_handoff lds_flag_handoff.hip ;  ./lds_flag_handoff [tiles] [mfma] [jitter]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int R = 8, TILE = 16384, AHEAD = 6;

__device__ __forceinline__ void dma2(const void* base, unsigned lds, unsigned off) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %3, %2\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "global_load_lds_dwordx4 %3, %2 offset:1024\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds), "s"(base), "v"(off)
        : "memory", "scc");
}

__device__ __forceinline__ unsigned lds_load(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int MODE>
__global__ void __launch_bounds__(512, 1) k(const unsigned char* src, size_t span_tiles, int tiles, int mfma, int jitter,
                                            float* out, long long* cycles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned* ready = reinterpret_cast<unsigned*>(smem + R * TILE);
    unsigned* consumed = ready + R;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int group = wave >> 2;
    const unsigned smem_base =
        __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem);
    if (threadIdx.x < 2 * R) ready[threadIdx.x] = 0;
    __syncthreads();
    size_t tsrc = (size_t)blockIdx.x * 7919u % span_tiles;
    auto next_src = [&]() { tsrc = (tsrc * 1664525u + 1013904223u) % span_tiles; };
    auto stage = [&](int t) {   // my 2 KiB of tile t
        dma2(src + tsrc * TILE + wave * 2048, smem_base + (t % R) * TILE + wave * 2048, lane * 16);
        next_src();
    };
    f32x16 acc, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = acc2[r] = 0.f;
    bf16x8 b;
#pragma unroll
    for (int r = 0; r < 8; ++r) b[r] = (__bf16)(0.001f * (lane + r));
    for (int t = 0; t < AHEAD; ++t) stage(t);
    unsigned seed = 12345u + 977u * group;
    unsigned pre_ready = 0, pre_cons = 0;
    const long long c0 = (long long)clock64();
    for (int t = 0; t < tiles; ++t) {
        // ---- producer half: slot of tile t + AHEAD was last read as tile t + AHEAD - R = t - 2
        const int tn = t + AHEAD, sn = tn % R;
        if (MODE != 0 && tn >= R) {
            const unsigned need = 8u * (unsigned)(tn / R);
            unsigned v = (MODE == 2) ? pre_cons : lds_load(&consumed[sn]);
            while (v < need) {
                __builtin_amdgcn_s_sleep(1);
                v = lds_load(&consumed[sn]);
            }
        }
        stage(tn);
        // my piece of tile t has landed when at most 2 * AHEAD DMA instructions of mine are still in flight
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        const int s = t % R;
        if (MODE == 0) {
            __syncthreads();
        } else {
            if (lane == 0) __hip_atomic_fetch_add(&ready[s], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            const unsigned need = 8u * (unsigned)(t / R + 1);
            unsigned v = (MODE == 2) ? pre_ready : lds_load(&ready[s]);
            while (v < need) {
                __builtin_amdgcn_s_sleep(1);
                v = lds_load(&ready[s]);
            }
            if (MODE == 2) {   // next step's two flags: requested now, consumed one step later
                pre_ready = lds_load(&ready[(t + 1) % R]);
                pre_cons = lds_load(&consumed[(t + 1 + AHEAD) % R]);
            }
        }
        // ---- consumer half: 16 fragment reads of the tile, `n` MFMAs
        seed = seed * 1664525u + 1013904223u;
        const int n = mfma + (jitter ? (int)((seed >> 16) % (unsigned)(2 * jitter + 1)) - jitter : 0);
        const unsigned char* tp = smem + s * TILE;
        for (int i = 0; i < n; i += 2) {      // two independent accumulators: no dependent-MFMA stall
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(tp + ((i & 15) * 1024 + lane * 16));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2, 0, 0, 0);
        }
        if (MODE != 0 && lane == 0)
            __hip_atomic_fetch_add(&consumed[s], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    const long long c1 = (long long)clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) sum += acc[r] + acc2[r];
    out[blockIdx.x * 512 + threadIdx.x] = sum;
    if (threadIdx.x == 0) cycles[blockIdx.x] = c1 - c0;
}

template <int MODE>
static void run(const char* name, const unsigned char* src, size_t span_tiles, int tiles, int mfma, int jitter, float* out,
                long long* cyc) {
    const size_t lds = R * TILE + 2 * R * sizeof(unsigned);
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), lds, 0, src, span_tiles, tiles, mfma, jitter, out, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), lds, 0, src, span_tiles, tiles, mfma, jitter, out, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < 256; ++i) mean += (double)h[i];
    mean /= 256;
    printf("{\"mode\": \"%s\", \"tiles\": %d, \"mfma_per_tile_per_wave\": %d, \"jitter\": %d, \"ms\": %.3f, "
           "\"us_per_tile\": %.4f, \"shader_clocks_per_tile\": %.1f, \"staged_GBps\": %.1f}\n",
           name, tiles, mfma, jitter, ms, ms * 1e3 / tiles, mean / tiles, 256.0 * tiles * TILE / (ms * 1e-3) / 1e9);
}

int main(int argc, char** argv) {
    const int tiles = argc > 1 ? atoi(argv[1]) : 4000;
    const int mfma = argc > 2 ? atoi(argv[2]) : 32;      // the LP kernel: 16 QK^T + 16 P.V MFMAs per wave and 64-key tile
    const int jitter = argc > 3 ? atoi(argv[3]) : 0;
    const size_t span_tiles = (size_t)(512u << 20) / TILE;   // 512 MB source: beyond L2 and the Infinity Cache
    unsigned char* src;
    float* out;
    long long* cyc;
    hipMalloc(&src, span_tiles * TILE);
    hipMemset(src, 1, span_tiles * TILE);
    hipMalloc(&out, 256 * 512 * sizeof(float));
    hipMalloc(&cyc, 256 * sizeof(long long));
    run<0>("s_barrier", src, span_tiles, tiles, mfma, jitter, out, cyc);
    run<1>("lds_flags_read_when_needed", src, span_tiles, tiles, mfma, jitter, out, cyc);
    run<2>("lds_flags_read_one_step_ahead", src, span_tiles, tiles, mfma, jitter, out, cyc);
    return 0;
}
