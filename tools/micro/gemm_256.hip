// Micro-benchmark (round 5): a hand-written 256 x 256 x 64 8-wave bf16 GEMM with LDS-DMA staging against
// hipBLASLt on the six GEMM shapes of a HunyuanVideo DiT block at M = 115 200 / 115 456 -- the same-box record of whether an
// own GEMM (the precondition of carrying RMSNorm + RoPE + pooling in the QKV GEMM's epilogue) can hold
// hipBLASLt's 0.53-0.60 of the MFMA peak under the 1400 W board cap.  NOT part of the library.
//
//   out[M, N] = x[M, K] . w[N, K]^T      (nn.Linear layout: both operands K-contiguous), bf16 in / out, fp32 accumulate
//
// Structure (cdna_hip_programming.md section 5, the "glds, 2 LDS buffers, BK = 64" row): workgroup = 8 waves (2 along M x 4
// along N... here 2 x 4 over (m, n) = wave tile 128 x 64), one workgroup per CU; v_mfma_f32_32x32x16_bf16 with A = W rows and
// B = X rows, so that a lane holds 4 CONSECUTIVE n of ONE token per register quad (8-byte stores, and the form a per-token
// epilogue would want); LDS tiles [256][64] bf16 with the 16-byte chunk index XOR-ed by (row >> 1) & 7 (conflict-free
// ds_read_b128 of 32 consecutive rows), the swizzle applied on the per-lane SOURCE address of global_load_lds_dwordx4; two
// LDS stages (128 KiB): the DMA of K-tile t + 1 is in flight while tile t is multiplied; workgroup ids remapped so that each
// XCD gets a contiguous range of tiles (4 m-tiles x all n-tiles at a time).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gemm_256 gemm_256.hip -lhipblaslt && ./gemm_256
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 256, BN = 256, BK = 64, THREADS = 512;
constexpr int TILE_BYTES = 256 * BK * 2;            // one operand tile: 32 KiB
constexpr int STAGE_BYTES = 2 * TILE_BYTES;         // X tile + W tile
constexpr int LDS_BYTES = 2 * STAGE_BYTES;          // 128 KiB

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)
#define LT(x)                                                              \
    do {                                                                   \
        hipblasStatus_t s_ = (x);                                          \
        if (s_ != HIPBLAS_STATUS_SUCCESS) {                                \
            fprintf(stderr, "%s: status %d (line %d)\n", #x, (int)s_, __LINE__); \
            exit(1);                                                       \
        }                                                                  \
    } while (0)

// one 1-KiB LDS-DMA piece: this wave's 8 rows [row0, row0 + 8) of an operand tile whose rows are `ld` elements apart in memory
__device__ __forceinline__ void stage8(const unsigned short* g, long long ld, int row0, int k0, unsigned char* lds_tile, int lane) {
    const int row = row0 + (lane >> 3), pchunk = lane & 7;
    const int c = pchunk ^ ((row >> 1) & 7);
    const unsigned short* src = g + (long long)row * ld + k0 + c * 8;
    __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(lds_tile + row0 * 128), 16, 0, 0);
}

__global__ void __launch_bounds__(THREADS, 2) gemm_256_kernel(const unsigned short* __restrict__ x, const unsigned short* __restrict__ w,
                                                             unsigned short* __restrict__ out, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;                 // wave tile: rows [128 wm, +128) of m, [64 wn, +64) of n
    const int mt_n = M / BM, nt_n = N / BN, nwg = mt_n * nt_n;
    // XCD-aware remap (bijective for any nwg): the hardware deals ids round robin over 8 XCDs
    const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    // tile order inside the range: groups of 4 m-tiles, inside a group n-tile major, m-tile minor
    const int per_group = 4 * nt_n, grp = wgid / per_group, in_g = wgid % per_group;
    const int g_m = (mt_n - grp * 4) < 4 ? (mt_n - grp * 4) : 4;
    const int mt = grp * 4 + in_g % g_m, nt = in_g / g_m;
    const unsigned short* xg = x + (long long)mt * BM * K;
    const unsigned short* wg = w + (long long)nt * BN * K;
    const int KT = K / BK;

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto stage = [&](int kt, int buf) {
        unsigned char* xs = smem + buf * STAGE_BYTES;
        unsigned char* ws = xs + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            stage8(xg, K, (i * 8 + wave) * 8, kt * BK, xs, lane);
            stage8(wg, K, (i * 8 + wave) * 8, kt * BK, ws, lane);
        }
    };
    // fragment addresses: 32 consecutive rows, logical 16-byte chunk = 2 ks + (lane >> 5)
    const int lr = lane & 31, hi = lane >> 5;
    int xoff[4], woff[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) xoff[j] = (wm * 128 + j * 32 + lr) * 128;
#pragma unroll
    for (int i = 0; i < 2; ++i) woff[i] = TILE_BYTES + (wn * 64 + i * 32 + lr) * 128;
    const int sw = (lr >> 1) & 7;          // rows of a fragment: (row >> 1) & 7 = (lr >> 1) & 7 (tile bases are multiples of 32)

    int cbs[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) cbs[ks] = ((2 * ks + hi) ^ sw) << 4;
    bf16x8 wf[2][2], xf[2][4];      // fragment double buffer: the reads of k-substep ks + 1 are in flight under the MFMAs of ks
    auto load_frags = [&](const unsigned char* base, int ks, int slot) {
#pragma unroll
        for (int i = 0; i < 2; ++i) wf[slot][i] = *reinterpret_cast<const bf16x8*>(base + woff[i] + cbs[ks]);
#pragma unroll
        for (int j = 0; j < 4; ++j) xf[slot][j] = *reinterpret_cast<const bf16x8*>(base + xoff[j] + cbs[ks]);
    };
    stage(0, 0);
    __syncthreads();
    load_frags(smem, 0, 0);
    for (int kt = 0; kt < KT; ++kt) {
        if (kt + 1 < KT) stage(kt + 1, (kt + 1) & 1);
        const unsigned char* base = smem + (kt & 1) * STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks < 3) load_frags(base, ks + 1, (ks + 1) & 1);
            if (ks == 3) {
                __syncthreads();          // K-tile kt + 1 has landed (vmcnt(0) + barrier); every wave is done reading tile kt
                if (kt + 1 < KT) load_frags(smem + ((kt + 1) & 1) * STAGE_BYTES, 0, 0);
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks & 1][i], xf[ks & 1][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
    }
    // epilogue: D[n][m]: lane -> token m = lane & 31, register r -> n = (r & 3) + 8 (r >> 2) + 4 hi: 4 consecutive n per quad
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long long m = (long long)mt * BM + wm * 128 + j * 32 + lr;
            const int n0 = nt * BN + wn * 64 + i * 32 + 4 * hi;
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const f32x4 v = {acc[i][j][rq * 4 + 0], acc[i][j][rq * 4 + 1], acc[i][j][rq * 4 + 2], acc[i][j][rq * 4 + 3]};
                const bf16x4 b = __builtin_convertvector(v, bf16x4);
                *reinterpret_cast<bf16x4*>(out + m * N + n0 + rq * 8) = b;
            }
        }
}

struct LtGemm {
    hipblasLtHandle_t h;
    hipblasLtMatmulDesc_t desc;
    hipblasLtMatrixLayout_t A, B, C;
    hipblasLtMatmulAlgo_t algo;
    void* ws;
    size_t ws_bytes = 64 << 20;
    void init(int M, int N, int K) {
        LT(hipblasLtCreate(&h));
        LT(hipblasLtMatmulDescCreate(&desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
        const hipblasOperation_t ta = HIPBLAS_OP_T, tb = HIPBLAS_OP_N;
        LT(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta)));
        LT(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb)));
        LT(hipblasLtMatrixLayoutCreate(&A, HIP_R_16BF, K, N, K));     // W' [K, N], op = T   (csrc/gemm.cpp's convention)
        LT(hipblasLtMatrixLayoutCreate(&B, HIP_R_16BF, K, M, K));     // X' [K, M]
        LT(hipblasLtMatrixLayoutCreate(&C, HIP_R_16BF, N, M, N));     // out' [N, M]
        hipblasLtMatmulPreference_t pref;
        LT(hipblasLtMatmulPreferenceCreate(&pref));
        LT(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws_bytes, sizeof(ws_bytes)));
        hipblasLtMatmulHeuristicResult_t hr[1];
        int found = 0;
        LT(hipblasLtMatmulAlgoGetHeuristic(h, desc, A, B, C, C, pref, 1, hr, &found));
        if (found < 1) { fprintf(stderr, "no hipBLASLt solution\n"); exit(1); }
        algo = hr[0].algo;
        CHECK(hipMalloc(&ws, ws_bytes));
        hipblasLtMatmulPreferenceDestroy(pref);
    }
    void run(const void* x, const void* w, void* out) {
        const float one = 1.f, zero = 0.f;
        LT(hipblasLtMatmul(h, desc, &one, w, A, x, B, &zero, out, C, out, C, &algo, ws, ws_bytes, 0));
    }
};

__global__ void fill(unsigned short* p, size_t n, unsigned seed) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) {
        unsigned h = (unsigned)(i * 2654435761u) ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        const float f = ((h & 0xffff) / 32768.0f - 1.0f);          // uniform [-1, 1): random operands (guide 5.4 rule 25)
        p[i] = (unsigned short)(__float_as_uint(f) >> 16);
    }
}
__global__ void diff(const unsigned short* a, const unsigned short* b, size_t n, float* out) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    float mx = 0.f, ref = 0.f;
    for (; i < n; i += stride) {
        const float x = __uint_as_float((unsigned)a[i] << 16), y = __uint_as_float((unsigned)b[i] << 16);
        mx = fmaxf(mx, fabsf(x - y));
        ref = fmaxf(ref, fabsf(y));
    }
    atomicMax((int*)out, __float_as_int(mx));
    atomicMax((int*)out + 1, __float_as_int(ref));
}

int main(int argc, char** argv) {
    struct Shape { const char* name; int M, N, K; };
    const Shape shapes[] = {{"qkv (double block, img)", 115200, 9216, 3072}, {"proj", 115200, 3072, 3072},
                            {"fc1", 115200, 12288, 3072}, {"fc2", 115200, 3072, 12288},
                            {"linear1 MLP half", 115456, 12288, 3072}, {"linear2", 115456, 3072, 15360}};
    const int reps = argc > 1 ? atoi(argv[1]) : 12;
    CHECK(hipFuncSetAttribute((const void*)gemm_256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    for (const Shape& s : shapes) {
        unsigned short *x, *w, *o1, *o2;
        float* d;
        CHECK(hipMalloc(&x, (size_t)s.M * s.K * 2));
        CHECK(hipMalloc(&w, (size_t)s.N * s.K * 2));
        CHECK(hipMalloc(&o1, (size_t)s.M * s.N * 2));
        CHECK(hipMalloc(&o2, (size_t)s.M * s.N * 2));
        CHECK(hipMalloc(&d, 8));
        CHECK(hipMemset(d, 0, 8));
        hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, x, (size_t)s.M * s.K, 1u);
        hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, w, (size_t)s.N * s.K, 2u);
        LtGemm lt;
        lt.init(s.M, s.N, s.K);
        const int grid = (s.M / BM) * (s.N / BN);
        auto own = [&]() { hipLaunchKernelGGL(gemm_256_kernel, dim3(grid), dim3(THREADS), LDS_BYTES, 0, x, w, o1, s.M, s.N, s.K); };
        own();
        lt.run(x, w, o2);
        hipLaunchKernelGGL(diff, dim3(2048), dim3(256), 0, 0, o1, o2, (size_t)s.M * s.N, d);
        float hd[2];
        CHECK(hipMemcpy(hd, d, 8, hipMemcpyDeviceToHost));
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        float ms_own = 0, ms_lt = 0;
        for (int pass = 0; pass < 2; ++pass) {          // interleaved: own, hipBLASLt, own, hipBLASLt (power steady state for both)
            for (int w_ = 0; w_ < 2; ++w_) own();
            hipEventRecord(e0);
            for (int r_ = 0; r_ < reps; ++r_) own();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float t;
            hipEventElapsedTime(&t, e0, e1);
            ms_own = t / reps;
            for (int w_ = 0; w_ < 2; ++w_) lt.run(x, w, o2);
            hipEventRecord(e0);
            for (int r_ = 0; r_ < reps; ++r_) lt.run(x, w, o2);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&t, e0, e1);
            ms_lt = t / reps;
        }
        const double fl = 2.0 * s.M * s.N * s.K;
        printf("{\"shape\": \"%s\", \"M\": %d, \"N\": %d, \"K\": %d, \"own_ms\": %.3f, \"own_TFLOPs\": %.1f, \"own_frac\": %.4f, "
               "\"hipblaslt_ms\": %.3f, \"hipblaslt_TFLOPs\": %.1f, \"hipblaslt_frac\": %.4f, \"max_abs_diff\": %.4g, \"max_abs_ref\": %.4g}\n",
               s.name, s.M, s.N, s.K, ms_own, fl / ms_own / 1e9, fl / ms_own / 1e9 / 2500, ms_lt, fl / ms_lt / 1e9, fl / ms_lt / 1e9 / 2500,
               hd[0], hd[1]);
        fflush(stdout);
        hipFree(x); hipFree(w); hipFree(o1); hipFree(o2); hipFree(d); hipFree(lt.ws);
    }
    return 0;
}
