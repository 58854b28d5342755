// Dense linear layers of the DiT blocks with the element-wise work around them folded into the GEMM's own epilogue
// ("fusing AttenCarve's neighbours into the DiT block ... adaLN modulate / apply_gate").
// Host code only: the GEMM is hipBLASLt's (a plain library GEMM; nothing here is a hand-written matrix kernel), what this
// file adds is the choice of epilogue / pointer mode / leading dimensions that torch's front-end does not expose:
//   act = GELU   out = gelu_tanh(x W^T + b) written with an arbitrary row stride  -> the MLP half of the single-stream
//                blocks' linear1 lands in linear2's concat buffer directly: the separate 5.7 GB GELU pass is gone
//                (models_mul_block_gc_ha_multigpu.py:404-406, 498-499)
//   gate / res   out = res + gate * (x W^T) + b'   (per-channel gate as hipBLASLt's alpha VECTOR, the residual as the
//                C matrix with beta = 1)           -> apply_gate + residual add of proj / fc2 / linear2
//                (models_mul_block_gc_ha_multigpu.py:297-315, 500; modulate_layers.py:53-68) in the GEMM epilogue
// Row-major [M,K] x [N,K]^T is handed to the column-major library as D'[N,M] = op_T(W'[K,N]) * X'[K,M], the same "TN"
// call torch makes for nn.Linear.
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <hipblaslt/hipblaslt-ext.hpp>

#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>
#include <utility>

#include "../../include/jenga_amd.h"

namespace jenga {
void set_error(const char* fmt, ...);
}
using jenga::set_error;

namespace {

struct Plan {
    hipblasLtMatmulDesc_t desc = nullptr;
    hipblasLtMatrixLayout_t A = nullptr, B = nullptr, C = nullptr, D = nullptr;
    hipblasLtMatmulAlgo_t algo;
    size_t workspace = 0;
    int choice = 0;      // index of the algorithm in the heuristic's list ...
    int want = 1;        // ... that was requested with this many candidates (the list is not prefix-stable across requests)
    int sol = -1;        // hipBLASLt's solution index of the algorithm (hipblaslt_ext::getIndexFromAlgo), -1 unknown
    void destroy() {
        if (desc) hipblasLtMatmulDescDestroy(desc);
        if (A) hipblasLtMatrixLayoutDestroy(A);
        if (B) hipblasLtMatrixLayoutDestroy(B);
        if (C) hipblasLtMatrixLayoutDestroy(C);
        if (D) hipblasLtMatrixLayoutDestroy(D);
        desc = nullptr;
        A = B = C = D = nullptr;
    }
};
// a plan under construction: every early return (LT_TRY) releases what was created so far
struct PlanGuard {
    Plan p;
    hipblasLtMatmulPreference_t pref = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool keep = false;
    ~PlanGuard() {
        if (pref) hipblasLtMatmulPreferenceDestroy(pref);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        if (!keep) p.destroy();
    }
};
// (M, N, K, x stride, w stride, ldc, out stride, epilogue, mode | has-res, dtype, workspace bytes) -- the device is NOT
// part of the shape key: a choice made on one rank is valid on every rank of the node
using Shape = std::tuple<long long, long long, long long, long long, long long, long long, long long, int, int, int, long long>;
using Key = std::pair<int, Shape>;
constexpr int SHAPE_FIELDS = 11;

std::mutex g_mu;
std::map<int, hipblasLtHandle_t> g_handles;
std::map<Key, Plan> g_plans;
struct Forced {
    int choice, want, sol;
};
std::map<Shape, Forced> g_forced;  // imported choices (jenga_linear_import_choices): no timing, take this algorithm
int g_import_mismatches = 0;       // plans rebuilt from an import whose solution index is not the exported one

#define LT_TRY(call)                                                             \
    do {                                                                         \
        hipblasStatus_t st_ = (call);                                            \
        if (st_ != HIPBLAS_STATUS_SUCCESS) {                                     \
            set_error("jenga_linear: %s failed with hipBLAS status %d", #call, (int)st_); \
            return JENGA_ELAUNCH;                                                \
        }                                                                        \
    } while (0)

void shape_to_array(const Shape& s, int64_t* o) {
    o[0] = std::get<0>(s); o[1] = std::get<1>(s); o[2] = std::get<2>(s); o[3] = std::get<3>(s); o[4] = std::get<4>(s);
    o[5] = std::get<5>(s); o[6] = std::get<6>(s); o[7] = std::get<7>(s); o[8] = std::get<8>(s); o[9] = std::get<9>(s);
    o[10] = std::get<10>(s);
}

}  // namespace

extern "C" int jenga_linear(void* stream, const void* x, const void* w, const void* bias, const void* res,
                            const float* gate, void* out, int64_t M, int64_t N, int64_t K, int64_t x_row_stride,
                            int64_t w_row_stride, int64_t res_row_stride, int64_t out_row_stride, int act,
                            void* workspace, int64_t workspace_bytes, int dtype) {
    // act carries the activation in its low byte and JENGA_BIAS_F32 as a flag: the bias vector is float32 (the gated
    // bias gate * b of proj / fc2 / linear2 stays unrounded on its way into the fp32 accumulator)
    const bool bias32 = (act & JENGA_BIAS_F32) != 0;
    // JENGA_OUT_F32 (round 6): the C (residual) and D (output) matrices are float32 -- the Wan blocks' residual stream
    // (wan/modules/model_mul.py:334-341: x fp32 + y * e): gate * (x W^T) + b' + res lands in the fp32 stream straight from the
    // fp32 accumulator, the separate gate + residual pass over 3.9 GB per call is gone
    const bool out32 = (act & JENGA_OUT_F32) != 0;
    act &= 0xff;
    if (!x || !w || !out || M < 0 || N <= 0 || K <= 0 || x_row_stride < K || w_row_stride < K || out_row_stride < N ||
        (res && res_row_stride < N) || (act != JENGA_ACT_NONE && act != JENGA_ACT_GELU_TANH) || workspace_bytes < 0 ||
        (workspace_bytes > 0 && !workspace)) {
        set_error("jenga_linear: bad arguments (row strides must cover the rows; act in {0, 1})");
        return JENGA_EINVAL;
    }
    if (out32 && act != JENGA_ACT_NONE) {
        set_error("jenga_linear: JENGA_OUT_F32 is for the gate / residual form (no activation epilogue)");
        return JENGA_EUNSUPPORTED;
    }
    if (act != JENGA_ACT_NONE && (res || gate)) {
        set_error("jenga_linear: the activation epilogue cannot be combined with gate / residual");
        return JENGA_EUNSUPPORTED;
    }
    if (dtype != JENGA_BF16 && dtype != JENGA_FP16) {
        set_error("jenga_linear: dtype must be bf16 or fp16");
        return JENGA_EUNSUPPORTED;
    }
    if (M == 0) return JENGA_OK;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        set_error("jenga_linear: no current HIP device");
        return JENGA_ELAUNCH;
    }
    std::lock_guard<std::mutex> lock(g_mu);
    hipblasLtHandle_t& handle = g_handles[dev];
    if (!handle) LT_TRY(hipblasLtCreate(&handle));
    const hipDataType dt = dtype == JENGA_BF16 ? HIP_R_16BF : HIP_R_16F;
    const int epi = (act == JENGA_ACT_GELU_TANH) ? (bias ? HIPBLASLT_EPILOGUE_GELU_BIAS : HIPBLASLT_EPILOGUE_GELU)
                                                 : (bias ? HIPBLASLT_EPILOGUE_BIAS : HIPBLASLT_EPILOGUE_DEFAULT);
    const int mode = gate ? HIPBLASLT_POINTER_MODE_ALPHA_DEVICE_VECTOR_BETA_HOST : HIPBLASLT_POINTER_MODE_HOST;
    const long long ldc = res ? res_row_stride : out_row_stride;
    // the workspace size is part of the key: an algorithm chosen with a 64 MiB workspace must not be replayed for a call
    // that brings a smaller one
    const Shape shape{(long long)M, (long long)N, (long long)K, (long long)x_row_stride, (long long)w_row_stride, ldc,
                      (long long)out_row_stride, epi, mode | (res ? 16 : 0) | (bias32 ? 32 : 0) | (out32 ? 64 : 0), dtype,
                      (long long)workspace_bytes};
    const Key key{dev, shape};
    auto it = g_plans.find(key);
    if (it == g_plans.end()) {
        PlanGuard g;
        Plan& p = g.p;
        LT_TRY(hipblasLtMatmulDescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
        const hipblasOperation_t ta = HIPBLAS_OP_T, tb = HIPBLAS_OP_N;
        LT_TRY(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta)));
        LT_TRY(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb)));
        const uint32_t e32 = (uint32_t)epi;
        LT_TRY(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &e32, sizeof(e32)));
        if (bias) {
            const int32_t bt = (int32_t)(bias32 ? HIP_R_32F : dt);
            LT_TRY(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt)));
            LT_TRY(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)));
        }
        const int32_t pm = (int32_t)mode;
        LT_TRY(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_POINTER_MODE, &pm, sizeof(pm)));
        LT_TRY(hipblasLtMatrixLayoutCreate(&p.A, dt, (uint64_t)K, (uint64_t)N, w_row_stride));    // W' [K,N], op = T
        LT_TRY(hipblasLtMatrixLayoutCreate(&p.B, dt, (uint64_t)K, (uint64_t)M, x_row_stride));    // X' [K,M]
        const hipDataType cdt = out32 ? HIP_R_32F : dt;
        LT_TRY(hipblasLtMatrixLayoutCreate(&p.C, cdt, (uint64_t)N, (uint64_t)M, ldc));             // res' [N,M]
        LT_TRY(hipblasLtMatrixLayoutCreate(&p.D, cdt, (uint64_t)N, (uint64_t)M, out_row_stride));  // out' [N,M]
        LT_TRY(hipblasLtMatmulPreferenceCreate(&g.pref));
        const uint64_t ws = (uint64_t)workspace_bytes;
        LT_TRY(hipblasLtMatmulPreferenceSetAttribute(g.pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws, sizeof(ws)));
        // JENGA_GEMM_CANDIDATES=k (default 1 = the library's first pick, no timing): time the heuristic's first k
        // solutions on this call's operands once, when the shape is first met, and keep the fastest -- for warm-up
        // passes only: it launches the GEMM several times and synchronises the stream, so it is refused while the stream
        // is being captured into a graph.  An imported choice (jenga_linear_import_choices) wins over both.
        int want = 1;
        if (const char* e = getenv("JENGA_GEMM_CANDIDATES")) want = atoi(e);
        if (want < 1) want = 1;
        if (want > 32) want = 32;
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing((hipStream_t)stream, &cap) != hipSuccess) {
            (void)hipGetLastError();
            cap = hipStreamCaptureStatusNone;
        }
        if (cap != hipStreamCaptureStatusNone) want = 1;
        // An imported choice is (list index, the candidate count the EXPORTER asked the heuristic for, solution index): the
        // list is re-queried with the exporter's count -- hipBLASLt does not promise that a shorter request returns a prefix
        // of a longer one -- and the algorithm is then looked up by its solution index; the list position is the fallback.
        int forced = -1, forced_sol = -1;
        auto fi = g_forced.find(shape);
        if (fi != g_forced.end()) {
            forced = fi->second.choice;
            forced_sol = fi->second.sol;
            want = fi->second.want < forced + 1 ? forced + 1 : fi->second.want;
            if (want > 32) want = 32;
        }
        const int asked = want;
        hipblasLtMatmulHeuristicResult_t hr[32];
        int found = 0;
        const hipblasStatus_t hs =
            hipblasLtMatmulAlgoGetHeuristic(handle, p.desc, p.A, p.B, p.C, p.D, g.pref, want, hr, &found);
        if (hs != HIPBLAS_STATUS_SUCCESS || found < 1) {
            set_error("jenga_linear: hipBLASLt has no solution for M=%lld N=%lld K=%lld epilogue=%d mode=%d (status %d)",
                      (long long)M, (long long)N, (long long)K, epi, mode, (int)hs);
            return JENGA_EUNSUPPORTED;
        }
        int best = 0;
        if (forced >= 0) {
            int by_sol = -1;
            if (forced_sol >= 0)
                for (int i = 0; i < found && by_sol < 0; ++i)
                    if (hr[i].state == HIPBLAS_STATUS_SUCCESS && hr[i].workspaceSize <= (size_t)workspace_bytes &&
                        hipblaslt_ext::getIndexFromAlgo(hr[i].algo) == forced_sol)
                        by_sol = i;
            if (by_sol >= 0)
                best = by_sol;
            else if (forced < found && hr[forced].state == HIPBLAS_STATUS_SUCCESS &&
                     hr[forced].workspaceSize <= (size_t)workspace_bytes)
                best = forced;       // (a list that came out shorter here than on the exporting rank: first pick)
            if (forced_sol >= 0 && hipblaslt_ext::getIndexFromAlgo(hr[best].algo) != forced_sol) ++g_import_mismatches;
        } else if (found > 1 && res != out) {   // (in place, every extra launch would add the residual once more)
            const float one_ = 1.0f, zero_ = 0.0f;
            const void* alpha_ = gate ? (const void*)gate : (const void*)&one_;
            const float* beta_ = res ? &one_ : &zero_;
            const void* c_ = res ? res : out;
            if (hipEventCreate(&g.e0) != hipSuccess || hipEventCreate(&g.e1) != hipSuccess) {
                set_error("jenga_linear: hipEventCreate failed while timing candidates");
                return JENGA_ELAUNCH;
            }
            float best_ms = 1e30f;
            for (int i = 0; i < found; ++i) {
                if (hr[i].state != HIPBLAS_STATUS_SUCCESS || hr[i].workspaceSize > (size_t)workspace_bytes) continue;
                bool ok = true;
                for (int rep = 0; rep < 5 && ok; ++rep) {      // 2 warm-up launches, 3 timed
                    if (rep == 2) ok = hipEventRecord(g.e0, (hipStream_t)stream) == hipSuccess;
                    ok = ok && hipblasLtMatmul(handle, p.desc, alpha_, w, p.A, x, p.B, beta_, c_, p.C, out, p.D,
                                               &hr[i].algo, workspace, (size_t)workspace_bytes,
                                               (hipStream_t)stream) == HIPBLAS_STATUS_SUCCESS;
                }
                if (hipEventRecord(g.e1, (hipStream_t)stream) != hipSuccess) ok = false;
                if (hipEventSynchronize(g.e1) != hipSuccess || !ok) continue;
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, g.e0, g.e1) != hipSuccess) continue;
                if (ms < best_ms) {
                    best_ms = ms;
                    best = i;
                }
            }
        }
        p.algo = hr[best].algo;
        p.workspace = hr[best].workspaceSize;
        p.choice = best;
        p.want = asked;
        p.sol = hipblaslt_ext::getIndexFromAlgo(p.algo);
        it = g_plans.emplace(key, p).first;
        g.keep = true;
    }
    Plan& p = it->second;
    if ((size_t)workspace_bytes < p.workspace) {     // (cannot happen with the size in the key; kept as the contract)
        set_error("jenga_linear: the plan needs %zu workspace bytes, the call brought %lld", p.workspace,
                  (long long)workspace_bytes);
        return JENGA_EINVAL;
    }
    if (bias)   // (the pointer is per call; the descriptor is shared under the lock)
        LT_TRY(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)));
    const float one = 1.0f, zero = 0.0f;
    const void* alpha = gate ? (const void*)gate : (const void*)&one;
    const float* beta = res ? &one : &zero;
    const void* c = res ? res : out;
    LT_TRY(hipblasLtMatmul(handle, p.desc, alpha, w, p.A, x, p.B, beta, c, p.C, out, p.D, &p.algo, workspace,
                           (size_t)workspace_bytes, (hipStream_t)stream));
    return JENGA_OK;
}

// Algorithm choices across the ranks of a job: every rank may time candidates on its own (JENGA_GEMM_CANDIDATES), but
// the replicated text stream must see the SAME arithmetic on every rank, so rank 0 exports its choices -- records of
// 12 int64: the 11 shape-key fields + (index in the heuristic's list | the candidate count that list was requested with << 8
// | (hipBLASLt solution index + 1) << 16) -- and every rank imports them (which drops the
// rank's own plans for those shapes; the next call rebuilds them from the imported index, no timing).
extern "C" int64_t jenga_linear_export_choices(int64_t* records, int64_t capacity) {
    std::lock_guard<std::mutex> lock(g_mu);
    int dev = 0;
    (void)hipGetDevice(&dev);
    int64_t n = 0;
    for (const auto& kv : g_plans) {
        if (kv.first.first != dev) continue;
        if (records && n < capacity) {
            shape_to_array(kv.first.second, records + n * (SHAPE_FIELDS + 1));
            // choice | requested candidate count << 8 | (solution index + 1) << 16   (0 in the upper part: unknown)
            records[n * (SHAPE_FIELDS + 1) + SHAPE_FIELDS] =
                (int64_t)kv.second.choice | ((int64_t)kv.second.want << 8) | ((int64_t)(kv.second.sol + 1) << 16);
        }
        ++n;
    }
    return n;
}

extern "C" int jenga_linear_import_choices(const int64_t* records, int64_t n) {
    if (n < 0 || (n > 0 && !records)) {
        set_error("jenga_linear_import_choices: bad arguments");
        return JENGA_EINVAL;
    }
    std::lock_guard<std::mutex> lock(g_mu);
    for (int64_t i = 0; i < n; ++i) {
        const int64_t* r = records + i * (SHAPE_FIELDS + 1);
        const int64_t packed = r[SHAPE_FIELDS];
        const int choice = (int)(packed & 0xff), want_ = (int)((packed >> 8) & 0xff), sol_ = (int)(packed >> 16) - 1;
        if (packed < 0 || choice >= 32 || want_ > 32) {
            set_error("jenga_linear_import_choices: record %lld has choice %d / requested count %d outside [0, 32]", (long long)i,
                      choice, want_);
            return JENGA_EINVAL;
        }
        const Shape s{r[0], r[1], r[2], r[3], r[4], r[5], r[6], (int)r[7], (int)r[8], (int)r[9], r[10]};
        g_forced[s] = Forced{choice, want_ < 1 ? choice + 1 : want_, sol_};
        for (auto it = g_plans.begin(); it != g_plans.end();) {
            if (it->first.second == s) {
                it->second.destroy();
                it = g_plans.erase(it);
            } else {
                ++it;
            }
        }
    }
    return JENGA_OK;
}

extern "C" int64_t jenga_linear_import_mismatches(void) {
    std::lock_guard<std::mutex> lock(g_mu);
    return g_import_mismatches;
}
