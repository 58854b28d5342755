// Reproducer (round 5): packed fp32 VALU arithmetic of one kernel gives wrong results while ANOTHER kernel's MFMAs run on the
// same GPU (gfx950, MI355X).  Stand-alone:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize tools/micro/pk_beside_mfma.hip -o pk_beside_mfma
// (with these flags the float2 forms stay packed and the scalar form stays scalar)
//   victim kernels: out[i] = f(in[i]) with f a chain of packed (float2) or scalar fp32 operations -- pure functions of the
//     input, so every launch must write the same bits;
//   load kernel: a loop of v_mfma_f32_16x16x32_bf16 (or 32x32x16) on registers (no memory traffic), launched back to back on
//     another stream.
// Prints, per victim form, launches whose output differs from the first launch: with the load / without it.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float f4v __attribute__((ext_vector_type(4)));
template <int M16>
__global__ void __launch_bounds__(256) mfma_load(float* sink, int iters) {
    bf8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x - i)); }
    if (M16) {          // v_mfma_f32_16x16x32_bf16: THE TRIGGER
        f4v acc = {0};
        for (int it = 0; it < iters; ++it) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, acc, 0, 0, 0);
        }
        if (acc[0] == 12345.678f) sink[0] = acc[1];     // keep the loop
    } else {            // v_mfma_f32_32x32x16_bf16: harmless
        f16v acc = {0};
        for (int it = 0; it < iters; ++it) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc, 0, 0, 0);
        }
        if (acc[0] == 12345.678f) sink[0] = acc[1];
    }
}

// FORM 0: packed mul + add on pairs built from unpacked bf16 halves (what the SLP vectoriser made of the LayerNorm kernel)
// FORM 1: the same arithmetic, scalar      FORM 2: packed fma only      FORM 3: packed mul only      FORM 4: packed add only
template <int FORM>
__global__ void __launch_bounds__(256) victim(const uint32_t* __restrict__ in, float* __restrict__ out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint4 r = reinterpret_cast<const uint4*>(in)[i];
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
    float s = 0.f;
    if (FORM == 1) {
        float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float lo = __uint_as_float(w[k] << 16), hi = __uint_as_float(w[k] & 0xffff0000u);
            acc0 = acc0 + lo * lo;
            acc1 = acc1 + hi * hi;
        }
        s = acc0 + acc1;
    } else {
        f2 acc = {0.f, 0.f};
        if (FORM == 3) acc = (f2){1.f, 1.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f2 v = {__uint_as_float(w[k] << 16), __uint_as_float(w[k] & 0xffff0000u)};
            if (FORM == 0) acc = acc + v * v;
            if (FORM == 2) acc = __builtin_elementwise_fma(v, v, acc);
            if (FORM == 3) acc = acc * v;
            if (FORM == 4) acc = acc + v;
            // 5: the pair minus the HIGH half of another pair broadcast to both lanes (op_sel:[0,1] -- the low result reads the
            //    high half; LayerNorm's x - mean); 6: minus the LOW half broadcast (op_sel_hi:[1,0])
            if (FORM == 5) { const f2 m = acc * (f2){0.5f, 0.25f}; acc = acc + (v - (f2){m.y, m.y}); }
            if (FORM == 6) { const f2 m = acc * (f2){0.5f, 0.25f}; acc = acc + (v - (f2){m.x, m.x}); }
            // 7: LayerNorm's instruction verbatim: (v.lo - m.hi, v.hi - m.hi)
            if (FORM == 7) {
                const f2 m = acc * (f2){0.5f, 0.25f};
                f2 d;
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(v), "v"(m));
                acc = acc + d;
            }
            // 9: the same operand selection without the negation: (v.lo + m.hi, v.hi + m.hi);  10: v_pk_mul_f32 with it
            if (FORM == 9) {
                const f2 m = acc * (f2){0.5f, 0.25f};
                f2 d;
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(v), "v"(m));
                acc = acc + d;
            }
            if (FORM == 10) {
                const f2 m = acc * (f2){0.5f, 0.25f} + (f2){1.f, 1.f};
                f2 d;
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(v), "v"(m));
                acc = acc + d;
            }
            // 11: v_pk_fma_f32 with the same selection on its second source;  12 / 13: the 16-bit packed add / mul with it
            if (FORM == 11) {
                const f2 m = acc * (f2){0.5f, 0.25f} + (f2){1.f, 1.f};
                f2 d;
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(d) : "v"(v), "v"(m), "v"(acc));
                acc = acc * (f2){0.5f, 0.5f} + d * (f2){0.25f, 0.25f};
            }
            if (FORM == 12 || FORM == 13) {
                uint32_t d;
                const uint32_t m = w[(k + 1) & 3];
                if (FORM == 12) asm volatile("v_pk_add_f16 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(w[k] & 0x3bff3bffu), "v"(m & 0x3bff3bffu));
                else asm volatile("v_pk_mul_f16 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(w[k] & 0x3bff3bffu), "v"(m & 0x3bff3bffu));
                acc = acc + (f2){(float)(d & 0xffffu), (float)(d >> 16)};
            }
            // round 6: 14 = the SWAP selection on the second source (low result reads the high half, high result the low half:
            //   op_sel:[0,1] op_sel_hi:[1,0] -- thousands of instances in torch's own kernels, profiles/r06_torch_pk_*);
            //   15 = the high half of the FIRST source broadcast (op_sel:[1,0]); 16 = v_pk_mul_f32 with the swap selection
            if (FORM == 14) {
                const f2 m = acc * (f2){0.5f, 0.25f};
                f2 d;
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(v), "v"(m));
                acc = acc + d;
            }
            if (FORM == 15) {
                const f2 m = acc * (f2){0.5f, 0.25f};
                f2 d;
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(d) : "v"(m), "v"(v));
                acc = acc + d;
            }
            if (FORM == 16) {
                const f2 m = acc * (f2){0.5f, 0.25f} + (f2){1.f, 1.f};
                f2 d;
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(v), "v"(m));
                acc = acc + d;
            }
            // round 6, narrowing the condition (torch kernels with these selections ran clean): is it the DISTANCE between the producer
            // of the source pair and the packed consumer?  17 = form 9 with 16 idle wait states between the two; 18 = form 9 with the
            // pair produced by two SCALAR multiplies; 19 = form 9 with the pair read back from memory (no VALU producer at all)
            if (FORM == 17) {
                f2 m = acc * (f2){0.5f, 0.25f};
                f2 d;
                asm volatile("s_nop 7\n\ts_nop 7\n\tv_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(v), "v"(m));
                acc = acc + d;
            }
            if (FORM == 18) {
                f2 m;
                asm volatile("v_mul_f32 %0, 0.5, %2\n\tv_mul_f32 %1, 0.25, %3" : "=&v"(m.x), "=&v"(m.y) : "v"(acc.x), "v"(acc.y));
                f2 d;
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(v), "v"(m));
                acc = acc + d;
            }
            if (FORM == 19) {
                const f2 m = {__uint_as_float(w[(k + 1) & 3] << 16), __uint_as_float(w[(k + 2) & 3] & 0xffff0000u)};
                f2 mm = m;
                asm volatile("s_nop 7\n\ts_nop 7" : "+v"(mm));      // (settled registers: nothing in flight)
                f2 d;
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(v), "v"(mm));
                acc = acc + d;
            }
            // is it the DESTINATION overlapping a source?  (an asm output without early-clobber may be given a source's registers.)
            // 20 = form 9 with a destination that overlaps NO source (early-clobber); 21 = destination IS the second source pair
            // (the one whose high half the low lane reads); 22 = destination IS the first source pair
            if (FORM == 20) {
                const f2 m = acc * (f2){0.5f, 0.25f};
                f2 d;
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=&v"(d) : "v"(v), "v"(m));
                acc = acc + d;
            }
            if (FORM == 21) {
                f2 m = acc * (f2){0.5f, 0.25f};
                asm volatile("v_pk_add_f32 %0, %1, %0 op_sel:[0,1]" : "+v"(m) : "v"(v));
                acc = acc + m;
            }
            if (FORM == 22) {
                const f2 m = acc * (f2){0.5f, 0.25f};
                f2 vv = v;
                asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1]" : "+v"(vv) : "v"(m));
                acc = acc + vv;
            }
            // 8: the mirror image: (v.lo - m.lo, v.hi - m.lo)
            if (FORM == 8) {
                const f2 m = acc * (f2){0.5f, 0.25f};
                f2 d;
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(v), "v"(m));
                acc = acc + d;
            }
        }
        s = acc.x + acc.y;
    }
    out[i] = s;
}

int main(int argc, char** argv) {
    const int n = 1 << 18, launches = argc > 1 ? atoi(argv[1]) : 20000;
    std::vector<uint32_t> h(n * 4);
    uint32_t x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (x & 0x7fff7fffu) | 0x3f003f00u; }   // bf16 pairs near 1
    uint32_t* din; float *dout, *dsink;
    CHECK(hipMalloc(&din, n * 16)); CHECK(hipMalloc(&dout, n * 4)); CHECK(hipMalloc(&dsink, 16));
    CHECK(hipMemcpy(din, h.data(), n * 16, hipMemcpyHostToDevice));
    hipStream_t sv, sl;
    CHECK(hipStreamCreate(&sv)); CHECK(hipStreamCreate(&sl));
    std::vector<float> first(n), cur(n);
    const char* names[23] = {"packed mul+add (SLP form)", "scalar", "packed fma", "packed mul", "packed add",
                            "packed sub of a broadcast HIGH half (op_sel:[0,1])", "packed sub of a broadcast LOW half (op_sel_hi:[1,0])",
                            "asm v_pk_add_f32 op_sel:[0,1] neg (x - mean, the LayerNorm instruction)", "asm v_pk_add_f32 op_sel_hi:[1,0] neg",
                            "asm v_pk_add_f32 op_sel:[0,1] (no neg)", "asm v_pk_mul_f32 op_sel:[0,1]",
                            "asm v_pk_fma_f32 op_sel:[0,1,0]", "asm v_pk_add_f16 op_sel:[0,1]", "asm v_pk_mul_f16 op_sel:[0,1]",
                            "asm v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0] (SWAP on src1)", "asm v_pk_add_f32 op_sel:[1,0] (HIGH half of src0 broadcast)",
                            "asm v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0] (SWAP on src1)",
                            "form 9 with 16 wait states between the pair's producer and the packed add",
                            "form 9 with the pair produced by two scalar v_mul_f32", "form 9 with the pair taken from settled registers (loaded data)",
                            "form 9, destination overlaps NO source (early-clobber)", "form 9, destination IS the second source pair",
                            "form 9, destination IS the first source pair"};
    for (int with_load = 2; with_load >= 0; --with_load) {      // 2: 16x16x32 MFMAs beside it, 1: 32x32x16, 0: nothing
        for (int form = 0; form < 23; ++form) {
            int bad = 0;
            for (int l = 0; l < launches; ++l) {
                if (with_load == 2 && (l % 4) == 0) hipLaunchKernelGGL(mfma_load<1>, dim3(2048), dim3(256), 0, sl, dsink, 4000);
                if (with_load == 1 && (l % 4) == 0) hipLaunchKernelGGL(mfma_load<0>, dim3(2048), dim3(256), 0, sl, dsink, 4000);
                switch (form) {
                    case 0: hipLaunchKernelGGL(victim<0>, dim3(n / 256), dim3(256), 0, sv, din, dout, n); break;
                    case 1: hipLaunchKernelGGL(victim<1>, dim3(n / 256), dim3(256), 0, sv, din, dout, n); break;
                    case 2: hipLaunchKernelGGL(victim<2>, dim3(n / 256), dim3(256), 0, sv, din, dout, n); break;
                    case 3: hipLaunchKernelGGL(victim<3>, dim3(n / 256), dim3(256), 0, sv, din, dout, n); break;
                    case 4: hipLaunchKernelGGL(victim<4>, dim3(n / 256), dim3(256), 0, sv, din, dout, n); break;
                    case 5: hipLaunchKernelGGL(victim<5>, dim3(n / 256), dim3(256), 0, sv, din, dout, n); break;
                    case 6: hipLaunchKernelGGL(victim<6>, dim3(n / 256), dim3(256), 0, sv, din, dout, n); break;
                    case 7: hipLaunchKernelGGL(victim<7>, dim3(n / 256), dim3(256), 0, sv, din, dout, n); break;
                    case 8: hipLaunchKernelGGL(victim<8>, dim3(n / 256), dim3(256), 0, sv, din, dout, n); break;
                    case 9: hipLaunchKernelGGL(victim<9>, dim3(n / 256), dim3(256), 0, sv, din, dout, n); break;
                    case 10: hipLaunchKernelGGL(victim<10>, dim3(n / 256), dim3(256), 0, sv, din, dout, n); break;
                    case 11: hipLaunchKernelGGL(victim<11>, dim3(n / 256), dim3(256), 0, sv, din, dout, n); break;
                    case 12: hipLaunchKernelGGL(victim<12>, dim3(n / 256), dim3(256), 0, sv, din, dout, n); break;
                    case 13: hipLaunchKernelGGL(victim<13>, dim3(n / 256), dim3(256), 0, sv, din, dout, n); break;
                    case 14: hipLaunchKernelGGL(victim<14>, dim3(n / 256), dim3(256), 0, sv, din, dout, n); break;
                    case 15: hipLaunchKernelGGL(victim<15>, dim3(n / 256), dim3(256), 0, sv, din, dout, n); break;
                    case 16: hipLaunchKernelGGL(victim<16>, dim3(n / 256), dim3(256), 0, sv, din, dout, n); break;
                    case 17: hipLaunchKernelGGL(victim<17>, dim3(n / 256), dim3(256), 0, sv, din, dout, n); break;
                    case 18: hipLaunchKernelGGL(victim<18>, dim3(n / 256), dim3(256), 0, sv, din, dout, n); break;
                    case 19: hipLaunchKernelGGL(victim<19>, dim3(n / 256), dim3(256), 0, sv, din, dout, n); break;
                    case 20: hipLaunchKernelGGL(victim<20>, dim3(n / 256), dim3(256), 0, sv, din, dout, n); break;
                    case 21: hipLaunchKernelGGL(victim<21>, dim3(n / 256), dim3(256), 0, sv, din, dout, n); break;
                    default: hipLaunchKernelGGL(victim<22>, dim3(n / 256), dim3(256), 0, sv, din, dout, n); break;
                }
                if (l == 0 || (l % 16) == 15) {          // check every 16th launch (and the first)
                    CHECK(hipMemcpyAsync(cur.data(), dout, n * 4, hipMemcpyDeviceToHost, sv));
                    CHECK(hipStreamSynchronize(sv));
                    if (l == 0) first = cur;
                    else if (memcmp(first.data(), cur.data(), n * 4) != 0) ++bad;
                }
            }
            CHECK(hipDeviceSynchronize());
            printf("{\"form\": \"%s\", \"mfma_load(2=16x16x32,1=32x32x16,0=none)\": %d, \"checked_launches\": %d, \"differing\": %d}\n", names[form], with_load,
                   launches / 16, bad);
            fflush(stdout);
        }
    }
    return 0;
}
