// Cross-XCD balancing of a launch (JENGA_ATTN_BALANCE), shared by the attention kernels: device side = per-XCD ticket
// queues, host side = the per-device counter sets.  Included once per translation unit; everything here has internal linkage,
// so every kernel file owns its counters.
//
// The hardware deals workgroup ids to the 8 XCDs round robin, so with one workgroup per work item every XCD gets the same
// number of items whatever its speed -- and the XCDs of one chip differ by several per cent.  Here a workgroup DRAWS its item:
// a ticket from the queue of the XCD it runs on (the same contiguous range, the same order as the static mapping), and once
// that queue is empty from the queue with the most items left.  The grid is oversubscribed so that a fast XCD has workgroups
// left to draw with; a workgroup that finds every queue empty exits.  Which workgroup computes an item does not enter the
// result: bit-identical to the static mapping.
//
// State this adds to the library (INTEGRATION.md "State"): LP_BAL_SETS x 8 ticket counters per device in a device global,
// and per device a mutex, LP_BAL_SETS events and the hand-out cursor.  Memory scopes the code relies on: the ticket draw is
// a device-scope returning atomic (performed at the memory side, so the 8 XCDs' L2s cannot each hold a private copy); the
// scan of the other queues is an `sc1` (agent-scope) load, and it is only a HINT -- a stale scan makes a workgroup try a
// queue that has just run dry, the draw itself decides, and the workgroup keeps trying while any queue still shows items.
#pragma once
#include <mutex>

#include "common.h"

namespace jenga {
namespace {

#define LP_BAL_SETS 64
__device__ int g_balance_ctr[LP_BAL_SETS][8];   // tickets drawn from each XCD's queue, one set per launch in flight (the
                                                // launcher hands the sets out in turn and zeroes one on the stream)

// The counter accesses are inline assembly WITHOUT a memory clobber, on purpose: an atomic the compiler can see counts as a
// possible write to everything the kernel loads afterwards, its uniform loads (launch order, kept count, sequence length,
// the list window) stop being scalar loads and come back through VGPRs, and with 256 VGPRs in use that costs the LP kernel's
// main loop one or two of its DMA offsets -- reloaded from scratch three times per 12 steps, each reload draining the DMA
// queue (measured: -2.8 % before any balancing gain).  Nothing else in the kernels reads or writes the counters.
typedef int lp_int4 __attribute__((ext_vector_type(4)));
// (not `volatile`, no memory clobber: to the compiler these are pure functions of their operands -- `seq` differs between
// any two calls of a workgroup so that they are never merged)
__device__ __forceinline__ int lp_ticket_add(int* p, int seq) {   // p: wave-uniform
    int old;
    const unsigned zero = 0;
    const int one = 1;
    asm("global_atomic_add %0, %1, %2, %3 sc0\n\ts_waitcnt vmcnt(0) ; draw %4" : "=v"(old) : "v"(zero), "v"(one), "s"(p), "s"(seq));
    return old;
}
// thread 0 draws (queue y, ticket t) -- own queue first, then the fullest other one, as long as any queue shows items --
// and the workgroup gets it through LDS as (y << 28 | t), or -1 when every queue is empty.  Queue z holds BH * nv(z) items,
// nv(z) = min(chunk, n_items - z * chunk) clamped at 0.
__device__ __forceinline__ int lp_draw_ticket(int* ctr, int BH, int n_items, int chunk, int x, int* lds) {
    if (threadIdx.x == 0) {
        auto qlen = [&](int z) {
            int nv = n_items - z * chunk;
            nv = nv < chunk ? nv : chunk;
            return nv > 0 ? BH * nv : 0;
        };
        int y = x, t = qlen(x);
        if (t > 0) t = lp_ticket_add(ctr + x, -1);
        if (t >= qlen(x)) {
            y = -1;
            // every failed attempt means another workgroup drew the ticket this one was after, so the loop ends after at
            // most (total tickets) rounds over the whole grid; 1 << 20 is a backstop, not a budget
#pragma nounroll
            for (int attempt = 0; attempt < (1 << 20) && y < 0; ++attempt) {
                lp_int4 c0, c1;
                const unsigned zero = 0;
                asm("global_load_dwordx4 %0, %2, %3 sc1\n\tglobal_load_dwordx4 %1, %2, %3 offset:16 sc1\n\t"
                    "s_waitcnt vmcnt(0) ; scan %4"
                    : "=&v"(c0), "=&v"(c1)
                    : "v"(zero), "s"(ctr), "s"(attempt));
                const int drawn[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
                int best = -1, left = 0;
#pragma unroll
                for (int z = 0; z < 8; ++z) {
                    const int l = qlen(z) - drawn[z];
                    if (l > left) { left = l; best = z; }
                }
                if (best < 0) break;
                best = __builtin_amdgcn_readfirstlane(best);     // (one lane is active: its value, in an SGPR for the asm)
                const int tt = lp_ticket_add(ctr + best, attempt);
                if (tt < qlen(best)) { y = best; t = tt; }
            }
        }
        lds[0] = y < 0 ? -1 : (y << 28 | t);
    }
    __syncthreads();
    const int ticket = __builtin_amdgcn_readfirstlane(lds[0]);
    __syncthreads();
    return ticket;
}

// ---- host side: the ticket counters of a launch.  LP_BAL_SETS sets per device, handed out in turn under a mutex (ranks
// simulated by threads launch concurrently on one device); a set is zeroed on the launch stream in front of the kernel, and
// an event recorded behind the kernel makes the NEXT user of the set -- LP_BAL_SETS launches later, possibly on another
// stream -- wait for it, so two launches in flight never share counters.  (A capturing stream never gets here: the launchers
// drop the flag, an event recorded inside a capture cannot order a set against launches outside.)
struct LpBalanceSlots {
    std::mutex mu;
    int* base = nullptr;
    hipEvent_t done[LP_BAL_SETS] = {};
    bool used[LP_BAL_SETS] = {};
    bool busy[LP_BAL_SETS] = {};
    unsigned next = 0;
};
LpBalanceSlots g_bal[64];

int lp_balance_acquire(hipStream_t stream) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
    LpBalanceSlots& S = g_bal[dev];
    std::lock_guard<std::mutex> lock(S.mu);
    if (!S.base && hipGetSymbolAddress((void**)&S.base, HIP_SYMBOL(g_balance_ctr)) != hipSuccess) {
        S.base = nullptr;
        return -1;
    }
    int k = -1;
    for (int tries = 0; tries < LP_BAL_SETS && k < 0; ++tries) {     // (a set between acquire and release belongs to another thread)
        const int c = (int)(S.next++ % LP_BAL_SETS);
        if (!S.busy[c]) k = c;
    }
    if (k < 0) return -1;
    if (!S.done[k] && hipEventCreateWithFlags(&S.done[k], hipEventDisableTiming) != hipSuccess) {
        S.done[k] = nullptr;
        return -1;
    }
    if (S.used[k] && hipStreamWaitEvent(stream, S.done[k], 0) != hipSuccess) return -1;
    if (hipMemsetAsync(S.base + 8 * k, 0, 8 * sizeof(int), stream) != hipSuccess) return -1;
    S.busy[k] = true;
    return k;
}

void lp_balance_release(int k, hipStream_t stream) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return;
    LpBalanceSlots& S = g_bal[dev];
    std::lock_guard<std::mutex> lock(S.mu);
    S.used[k] = hipEventRecord(S.done[k], stream) == hipSuccess;
    if (!S.used[k]) (void)hipStreamSynchronize(stream);   // (no event: the set must be idle before anybody reuses it)
    S.busy[k] = false;
}

// grid oversubscription of a balanced launch, per cent of the work items: JENGA_BALANCE_EXTRA_PCT, read ONCE per process
// (first launch), clamped to [0, 100]; the launchers add at least 8 workgroups whatever it says
inline int lp_balance_extra_pct() {
    static const int pct = [] {
        int v = 12;
        if (const char* ev = getenv("JENGA_BALANCE_EXTRA_PCT")) v = atoi(ev);
        return v < 0 ? 0 : (v > 100 ? 100 : v);
    }();
    return pct;
}

}  // namespace
}  // namespace jenga
